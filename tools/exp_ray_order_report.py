#!/usr/bin/env python3
"""Pairs the "order:" lines tools/exp_ray_order.py prints with the traversal-kernel dispatches of the same run (rocprofv3 --kernel-trace:
durations; a second run under --pmc TCC_HIT_sum TCC_MISS_sum: L2 hit rate) and prints one line per ray order.

  python tools/exp_ray_order_report.py <stdout log> <kernel-trace results.db> [<pmc results.db>] [--out profiles/r03x_ray_order_c4.txt]
"""
import argparse
import sqlite3


def trace_dispatches(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, duration from kernels order by start").fetchall()
    return [(n, d) for n, d in rows if "k_trace" in n]  # (k_route follows every k_trace_dyn<closest>; not counted)


def pmc_dispatches(db):
    cur = sqlite3.connect(db).cursor()
    per = {}
    for did, name, counter, value in cur.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name order by dispatch_id"):
        if "k_trace" in name:
            per.setdefault(did, {})[counter] = float(value)
    return [per[k] for k in sorted(per)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("kt_db")
    ap.add_argument("pmc_db", nargs="?")
    ap.add_argument("--out")
    a = ap.parse_args()
    labels = []
    for line in open(a.log):
        if line.startswith(("camera rays", "bounce-1", "bounce-2")):
            labels.append(("gen: " + line.split("hit fraction")[0].strip(), None))
        elif line.startswith("order:"):
            name, n = line[len("order:"):].rsplit(" ", 1)
            labels.append((name.strip(), int(n)))
    disp = trace_dispatches(a.kt_db)
    pmc = pmc_dispatches(a.pmc_db) if a.pmc_db else []
    out = [f"# ray order vs traversal time ({a.log}); {len(disp)} traversal dispatches, {len(labels)} labels",
           f"{'order':44s} {'rays':>9s} {'kernel':>14s} {'us':>10s} {'vs random':>10s} {'L2 hit':>8s}"]
    base = None
    for i, (name, n) in enumerate(labels):
        if i >= len(disp):
            break
        kn, dur = disp[i]
        us = dur / 1e3
        if name == "random mix":
            base = us
        l2 = ""
        if i < len(pmc) and (pmc[i].get("TCC_HIT_sum", 0) + pmc[i].get("TCC_MISS_sum", 0)) > 0:
            l2 = f"{pmc[i]['TCC_HIT_sum'] / (pmc[i]['TCC_HIT_sum'] + pmc[i]['TCC_MISS_sum']):.3f}"
        rel = f"{us / base - 1.0:+.1%}" if base and n else ""
        out.append(f"{name:44s} {(n if n else 0):9d} {kn.replace('gi::', '').split('<')[0]:>14s} {us:10.1f} {rel:>10s} {l2:>8s}")
    txt = "\n".join(out) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
