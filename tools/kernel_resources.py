#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

  python tools/kernel_resources.py gi_trace.hip [name filter] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gatling_amd import build as B  # noqa: E402


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    src = args[0]; flt = args[1] if len(args) > 1 else ""
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + B.KERNEL_FLAGS + extra + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, cwd=B.CSRC, stderr=subprocess.PIPE, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([^:]+): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    dem = subprocess.run(["c++filt"], input="\n".join(rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for name, d in zip(dem, rows.values()):
        short = name.replace("gi::", "").replace("void ", "").split("(")[0]
        if flt and flt not in short:
            continue
        print(f"{short:70s} VGPR {d.get('VGPRs', -1):4d} AGPR {d.get('AGPRs', 0):3d} spill {d.get('VGPRs Spill', 0):3d} scratch {d.get('ScratchSize [bytes/lane]', 0):4d} occ {d.get('Occupancy [waves/SIMD]', 0)} LDS {d.get('LDS Size [bytes/block]', 0)}")


if __name__ == "__main__":
    main()
