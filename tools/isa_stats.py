#!/usr/bin/env python3
"""Static instruction counts of the kernels in a device assembly file (hipcc --cuda-device-only -S): VALU / SALU / LDS / VMEM per kernel, VGPRs, occupancy.

  python tools/isa_stats.py /tmp/isa/gi_trace.s [name filter]      (tools/isa_stats.py --build gi_trace.hip [filter] compiles first)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(path, flt=""):
    names, cur, rows = [], None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); rows[cur] = dict(valu=0, salu=0, lds=0, vmem=0, cvt=0, vgpr=0, occ=0, sgpr=0, scratch=0); names.append(cur)
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("v_"):
            rows[cur]["valu"] += 1
            if t.startswith("v_cvt"):
                rows[cur]["cvt"] += 1
        elif t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_branch", "s_cbranch")):
            rows[cur]["salu"] += 1
        elif t.startswith("ds_"):
            rows[cur]["lds"] += 1
        elif t.startswith(("global_", "flat_", "buffer_", "scratch_")):
            rows[cur]["vmem"] += 1
        for key, pat in (("vgpr", r"; NumVgprs: (\d+)"), ("occ", r"; Occupancy: (\d+)"), ("sgpr", r"; TotalNumSgprs: (\d+)"), ("scratch", r"; ScratchSize: (\d+)")):
            m = re.match(pat, t)
            if m:
                rows[cur][key] = int(m.group(1))
        if t.startswith(".Lfunc_end"):
            pass
    dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        short = d.replace("gi::", "").replace("void ", "").split("(")[0]
        if flt and flt not in short:
            continue
        r = rows[n]
        print(f"{short:64s} VALU {r['valu']:5d} (cvt {r['cvt']:3d}) SALU {r['salu']:4d} LDS {r['lds']:3d} VMEM {r['vmem']:3d}  VGPR {r['vgpr']:3d} SGPR {r['sgpr']:3d} scratch {r['scratch']:4d} occ {r['occ']}")


def main():
    args = sys.argv[1:]
    if args and args[0] == "--build":
        sys.path.insert(0, ROOT)
        from gatling_amd import build as B
        src = args[1]; out = "/tmp/isa_" + os.path.basename(src) + ".s"
        extra = [a for a in args[2:] if a.startswith("-")]
        flt = next((a for a in args[2:] if not a.startswith("-")), "")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + B.KERNEL_FLAGS + extra + ["--cuda-device-only", "-S", src, "-o", out], cwd=B.CSRC,
                              stderr=subprocess.DEVNULL)
        stats(out, flt)
    else:
        stats(args[0], args[1] if len(args) > 1 else "")


if __name__ == "__main__":
    main()
