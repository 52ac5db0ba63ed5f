#!/usr/bin/env python3
"""How much of a kernel trace's busy time has two kernels running at once (rocprofv3 --kernel-trace result database, rocpd sqlite): the evidence that the two-stream
iterations of low-spp frames overlap the shadow launch of bounce i with the closest-hit launch of bounce i + 1 (gi_render.cpp "two streams").

  python tools/overlap_from_trace.py gpurun_out/prof/kt2/x_results.db [tag]     -> text on stdout (and profiles/<tag>_overlap.txt)"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("gi::", "").replace("void ", "").split("(")[0]


def main():
    path = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else None
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = cur.execute(f"select name, {start}, {end} from kernels order by {start}").fetchall()
    ev = []
    for n, s, e in rows:
        ev.append((s, 1, short(n))); ev.append((e, -1, short(n)))
    ev.sort()
    busy = both = 0
    live, last = [], None
    pairs = {}
    for t, d, n in ev:
        if last is not None and live:
            busy += t - last
            if len(live) >= 2:
                both += t - last
                key = " || ".join(sorted(set(live)))
                pairs[key] = pairs.get(key, 0) + (t - last)
        if d > 0: live.append(n)
        else: live.remove(n)
        last = t
    total = sum(e - s for _, s, e in rows)
    lines = [f"# kernel overlap in {os.path.basename(path)}: {len(rows)} dispatches",
             f"sum of kernel durations {total / 1e6:.3f} ms, time with a kernel running {busy / 1e6:.3f} ms, with TWO OR MORE running {both / 1e6:.3f} ms ({100.0 * both / max(busy, 1):.1f} % of the busy time)"]
    for k, v in sorted(pairs.items(), key=lambda kv: -kv[1])[:8]:
        lines.append(f"  {v / 1e6:8.3f} ms  {k}")
    txt = "\n".join(lines)
    print(txt)
    if tag:
        open(os.path.join(ROOT, "profiles", f"{tag}_overlap.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
