#!/usr/bin/env python3
"""Turns rocprofv3 result databases (rocpd sqlite, the default output of ROCm 7.2's rocprofv3) into the small text /
JSON summaries committed under profiles/.

  python tools/summarize_profile.py --kernel-trace gpurun_out/prof/kt/c2_results.db \
      --fetch gpurun_out/prof/fetch/c2_results.db --write gpurun_out/prof/write/c2_results.db --tag r01_c2 --spp-pmc 64

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section "HBM": FETCH_SIZE and WRITE_SIZE are collected in separate
--pmc passes (TCC slot limits); both are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests of wide coalesced
reads as 64 B, so it is doubled before comparing with byte counts; WRITE_SIZE is uncalibrated and used as reported.
"""
import argparse
import json
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("gi::", "")
    return name.split("(")[0].replace("void ", "")


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count),"
                       " max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    out = []
    for r in rows:
        out.append({"kernel": short(r[0]), "calls": r[1], "total_ms": r[2] / 1e6, "avg_us": r[3] / 1e3, "min_us": r[4] / 1e3, "max_us": r[5] / 1e3,
                    "pct": 100.0 * r[2] / total, "vgpr": r[6], "sgpr": r[7], "lds": r[8], "scratch": r[9], "grid": r[10], "block": r[11]})
    return out


def counter_sums(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {short(r[0]): {"dispatches": r[1], "kib": r[2]} for r in rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel-trace", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--spp", type=int, default=64)
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    ks = kernel_stats(args.kernel_trace)
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({args.tag}) {args.note}",
             f"{'kernel':34s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'grid':>9s}"]
    for k in ks:
        lines.append(f"{k['kernel']:34s} {k['calls']:7d} {k['total_ms']:10.3f} {k['avg_us']:9.3f} {k['min_us']:9.3f} {k['max_us']:9.3f} {k['pct']:6.2f} "
                     f"{k['vgpr']:5d} {k['sgpr']:5d} {k['lds']:7d} {k['scratch']:7d} {k['grid']:9d}")
    pmc = {}
    if args.fetch and args.write:
        f, w = counter_sums(args.fetch, "FETCH_SIZE"), counter_sums(args.write, "WRITE_SIZE")
        lines += ["", "# HBM traffic per launch from --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); fetch doubled per the gfx950 note",
                  f"{'kernel':34s} {'dispatches':>10s} {'fetch_KiB_raw':>14s} {'fetch_B/launch(x2)':>20s} {'write_B/launch':>16s} {'hbm_B/launch':>14s}"]
        for name in sorted(set(f) | set(w)):
            fd, wd = f.get(name, {"dispatches": 0, "kib": 0.0}), w.get(name, {"dispatches": 0, "kib": 0.0})
            n = max(1, fd["dispatches"])
            fb = 2.0 * fd["kib"] * 1024.0 / n
            wb = wd["kib"] * 1024.0 / max(1, wd["dispatches"])
            pmc[name] = {"dispatches": fd["dispatches"], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
            lines.append(f"{name:34s} {fd['dispatches']:10d} {fd['kib']:14.1f} {fb:20.1f} {wb:16.1f} {fb + wb:14.1f}")
    txt = os.path.join(ROOT, "profiles", f"{args.tag}_rocprofv3_summary.txt")
    open(txt, "w").write("\n".join(lines) + "\n")
    js = {"tag": args.tag, "kernels": ks, "pmc": pmc}
    if pmc:
        # the traversal stage's launches: k_trace<closest> (scene in LDS) or k_trace_dyn<closest> + its k_route pass
        tk = [v for k, v in pmc.items() if (k.startswith("k_trace<false") or k.startswith("k_trace_dyn<false")) and ", true," not in k[:30]]
        if tk:
            js["trace_bytes_per_launch"] = max(v["hbm_bytes_per_launch"] for v in tk) + (pmc["k_route"]["hbm_bytes_per_launch"] if "k_route" in pmc else 0.0)
    json.dump(js, open(os.path.join(ROOT, "profiles", f"{args.tag}_rocprofv3_summary.json"), "w"), indent=1)
    if pmc and "trace_bytes_per_launch" in js:
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        table = {}
        if os.path.exists(path):
            try:
                table = json.load(open(path))
                if "workload" in table:  # older single-entry layout
                    table = {table["workload"]: table}
            except Exception:
                table = {}
        table[args.workload] = {"tag": args.tag, "workload": args.workload, "spp": args.spp, "trace_bytes_per_launch": js["trace_bytes_per_launch"],
                                "note": "HBM bytes per traversal-stage launch = 2*FETCH_SIZE + WRITE_SIZE (separate --pmc passes); launches at this spp carry the "
                                        "same ray count per launch as the full-spp run once all pixels are active", "source": os.path.basename(txt)}
        json.dump(table, open(path, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
