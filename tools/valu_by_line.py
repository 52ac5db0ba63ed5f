#!/usr/bin/env python3
"""Where a kernel's static VALU instructions come from, by source line: compiles one kernel unit with line tables (-gline-tables-only, device only, -S) and adds up
the v_* instructions of one kernel per `.loc` (innermost inlined line).  The histogram that found round 6's "instruction diet" items.

  python tools/valu_by_line.py gi_shade.hip "k_shade<1u, false, false, true, true>" [top N = 40] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gatling_amd import build as B  # noqa: E402


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--"); extra = args[k + 1:]; args = args[:k]
    unit, want = args[0], args[1]
    top = int(args[2]) if len(args) > 2 else 40
    out = os.path.join(ROOT, "tools", "build", unit + ".lines.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + B.KERNEL_FLAGS + extra + ["-gline-tables-only", "--cuda-device-only", "-S", unit, "-o", out],
                          cwd=B.CSRC, stderr=subprocess.DEVNULL)
    files, per_line, per_file, cur, loc, total = {}, Counter(), Counter(), None, None, 0
    for line in open(out):
        m = re.match(r'\s*\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]+)"', line)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3)); continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            name = re.sub(r"^void gi::", "", name).split("(")[0]
            cur = name if name == want else None; continue
        if cur is None: continue
        if line.startswith(".Lfunc_end"): cur = None; continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m: loc = (int(m.group(1)), int(m.group(2))); continue
        t = line.strip()
        if t.startswith("v_") and loc:
            per_line[loc] += 1; per_file[loc[0]] += 1; total += 1
    if not total:
        raise SystemExit(f"kernel {want!r} not found in {unit}")
    print(f"# {want}: {total} static VALU instructions")
    for f, n in per_file.most_common():
        print(f"#   {files.get(f, f):<22s} {n:6d}  {100.0 * n / total:5.1f} %")
    src = {}
    for (f, ln), n in per_line.most_common(top):
        fn = files.get(f, str(f))
        if fn not in src:
            p = os.path.join(B.CSRC, fn)
            src[fn] = open(p).read().split("\n") if os.path.exists(p) else []
        text = src[fn][ln - 1].strip()[:110] if 0 < ln <= len(src[fn]) else ""
        print(f"{n:5d}  {fn}:{ln:<5d} {text}")


if __name__ == "__main__":
    main()
