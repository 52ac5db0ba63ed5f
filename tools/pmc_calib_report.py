#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of tools/build/pmc_calib (FETCH_SIZE, WRITE_SIZE) into per-pattern scale factors
(bytes the lanes asked for / KiB the counter reported * 1024) and writes profiles/pmc_calibration.json + a text table.

  python tools/pmc_calib_report.py <fetch_results.db> <write_results.db> [--tag r02]
"""
import argparse
import json
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GIB, COUNT = 2 << 30, 64 << 20
# launch order inside one repetition of pmc_calib's main(): (kernel, label, bytes asked for, reads?)
PATTERNS = [("calib_stream_read16", "stream read, 16 B/lane coalesced, 2 GiB", GIB, True),
            ("calib_gather64", "64-B slot gather, random over 2 GiB", COUNT * 64, True),
            ("calib_gather64", "64-B slot gather, random over 268 MB (4 Mi-slot pool)", COUNT * 64, True),
            ("calib_node80", "80-B node fetch (5 x dwordx4), random over 2 GiB", COUNT * 80, True),
            ("calib_node80", "80-B node fetch (5 x dwordx4), random over 20 MB (C3's BVH8)", COUNT * 80, True),
            ("calib_stream_write16", "stream write, 16 B/lane coalesced, 2 GiB", GIB, False),
            ("calib_scatter16", "16-B record scatter, random over 2 GiB", COUNT * 16, False),
            ("calib_write16_runs", "16-B records in runs of 64 (1 KiB), random over 2 GiB", COUNT * 16, False)]


def per_dispatch(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name=? group by dispatch_id order by dispatch_id", (counter,)).fetchall()
    return [(r[1].split("(")[0], float(r[2])) for r in rows if r[1].startswith("calib_")]  # (hipMemset's fill kernels are dispatches too)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--tag", default="r02")
    a = ap.parse_args()
    f, w = per_dispatch(a.fetch_db, "FETCH_SIZE"), per_dispatch(a.write_db, "WRITE_SIZE")
    n = len(PATTERNS)
    lines = [f"# FETCH_SIZE / WRITE_SIZE calibration on known byte counts ({a.tag}; tools/pmc_calib.hip, second repetition)",
             f"{'pattern':72s} {'asked_B':>14s} {'FETCH_KiB':>14s} {'WRITE_KiB':>14s} {'B per reported fetch B':>24s} {'B per reported write B':>24s}"]
    table = []
    for i, (k, label, asked, reads) in enumerate(PATTERNS):
        fi, wi = f[n + i] if len(f) >= 2 * n else f[i], w[n + i] if len(w) >= 2 * n else w[i]
        assert k in fi[0] and k in wi[0], (k, fi[0], wi[0])
        ff = asked / (fi[1] * 1024.0) if reads and fi[1] else None
        wf = asked / (wi[1] * 1024.0) if (not reads) and wi[1] else None
        table.append({"kernel": k, "pattern": label, "asked_bytes": asked, "fetch_kib": fi[1], "write_kib": wi[1], "fetch_factor": ff, "write_factor": wf})
        lines.append(f"{label:72s} {asked:14d} {fi[1]:14.1f} {wi[1]:14.1f} {(f'{ff:.3f}' if ff else '-'):>24s} {(f'{wf:.3f}' if wf else '-'):>24s}")
    # FETCH_SIZE on gfx950 = 64 B per fabric read request (counter_defs.yaml: TCC_BUBBLE, the 128-B request count, reads 0): wide
    # coalesced streams issue 128-B requests (reported / asked = 1/2), scattered <= 64-B sector fetches are reported at their size.
    # "fetch" = factor for coalesced streams, "fetch_scattered" = factor for 64-B sector gathers (slot gathers, BVH node / triangle fetches)
    out = {"tag": a.tag, "fetch": table[0]["fetch_factor"], "fetch_scattered": 1.0, "write": table[5]["write_factor"],
           "source": f"profiles/{a.tag}_pmc_calibration.txt (tools/pmc_calib.hip): stream read asked/reported = {table[0]['fetch_factor']:.3f}, 64-B gathers "
                     f"{table[1]['fetch_factor']:.3f}, 80-B node fetches {table[3]['fetch_factor']:.3f} (sector over-fetch is real traffic), stream write {table[5]['write_factor']:.3f}, "
                     f"scattered 16-B records {table[6]['write_factor']:.3f} (32-B write granule), 16-B records in 1 KiB runs {table[7]['write_factor']:.3f}", "patterns": table}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc_calibration.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_calibration.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
