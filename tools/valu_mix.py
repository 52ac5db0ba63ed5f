"""Instruction-class mix of the stage kernels (rocprofv3 --pmc, counters only): what share of a kernel's VALU instructions is fp32 arithmetic -- the class whose
issue ceiling (0.59 instructions per ns and SIMD, profiles/r03b_valu_issue_calibration.txt) applies to a stream that does not alternate with conversions / integer work.

  python tools/valu_mix.py c3 32     ->  per kernel: VALU instructions, shares of ADD / MUL / FMA / TRANS f32, INT32, CVT, and SALU / LDS / VMEM per VALU"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"],
          ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"],
          ["SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"],
          ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU_ADD_F16", "SQ_INSTS_VALU_FMA_F16"]]


def main():
    workload, spp = sys.argv[1], int(sys.argv[2])
    bench.PMC_PASSES = PASSES
    pmc, note = bench.pmc_live(workload, spp, timeout_s=300)
    print("# note:", note or "ok")
    if not pmc:
        return
    names = [c for g in PASSES for c in g]
    for k, v in sorted(pmc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0.0)):
        n = v.get("SQ_INSTS_VALU", 0.0)
        if n < 1e6:
            continue
        f32 = sum(v.get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"))
        other = sum(v.get(c, 0.0) for c in ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"))
        print(f"{k[:56]:56s} VALU {n:.3e}  f32 arithmetic {f32 / n:.3f}  int+cvt {other / n:.3f}  unclassified {1 - (f32 + other) / n:.3f}  | " +
              "  ".join(f"{c.replace('SQ_INSTS_', '')} {v[c] / n:.3f}" for c in names[1:] if c in v))


if __name__ == "__main__":
    main()
