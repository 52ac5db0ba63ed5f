#!/usr/bin/env python3
"""Turns tools/build/valu_calib's output (and, optionally, the rocprofv3 --pmc pass of `valu_calib --pmc`) into
profiles/valu_issue_calibration.{txt,json}.  bench.py reads `cycles_per_wave64_valu` from the json (VALU_CYCLES_PER_INST).

  python tools/valu_calib_report.py <valu_calib stdout> [<pmc results.db>] [--tag r03]
"""
import argparse
import json
import os
import re
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    rows, dev = [], {}
    for line in open(path):
        if line.startswith("VALU_CALIB_DEVICE"):
            dev = dict(re.findall(r'(\w+)=("[^"]*"|\S+)', line))
        if not line.startswith("VALU_CALIB "):
            continue
        kv = dict(re.findall(r'(\w+)=("[^"]*"|\S+)', line))
        rows.append({k: (v.strip('"') if v.startswith('"') else float(v)) for k, v in kv.items()})
    return dev, rows


def pmc_rows(db):
    """Per dispatch, in launch order: the counters of the k_valu kernels (valu_calib --pmc launches a warm-up before every timed one)."""
    cur = sqlite3.connect(db).cursor()
    out = {}
    for did, name, counter, value in cur.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name order by dispatch_id"):
        if "k_valu" not in name:
            continue
        out.setdefault(did, {"kernel": name})[counter] = float(value)
    return [out[k] for k in sorted(out)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("pmc_db", nargs="?")
    ap.add_argument("--pmc-log", help="stdout of the --pmc run (names the rows of pmc_db in launch order)")
    ap.add_argument("--tag", default="r03")
    a = ap.parse_args()
    dev, rows = parse(a.log)
    if not rows:
        raise SystemExit("no VALU_CALIB lines in " + a.log)
    lines = [f"# VALU issue-rate calibration ({a.tag}; tools/valu_calib.hip) on {dev.get('name', '?')} ({dev.get('cus', '?')} CUs)",
             "# cycles/instr/SIMD = median wave cycles (s_memtime) / (waves per SIMD x instructions per wave): clock independent",
             f"{'instruction':38s} {'waves/SIMD':>10s} {'cyc/instr/SIMD':>15s} {'instr/SIMD/ns (wall)':>21s} {'implied GHz':>12s} {'max waves per HW key':>21s}"]
    for r in rows:
        lines.append(f"{r['op']:38s} {int(r['waves_per_simd']):10d} {r['cycles_per_instr_per_simd']:15.4f} {r['wall_instr_per_simd_per_ns']:21.5f} "
                     f"{r['implied_clock_GHz_at_that_rate']:12.4f} {int(r['max_waves_sharing_hw_key']):21d}")
    # the constant: plain fp32 VALU at 8 waves/SIMD (issue-saturated)
    sat = {r["op"]: r["cycles_per_instr_per_simd"] for r in rows if int(r["waves_per_simd"]) == 8}
    rate = {r["op"]: r["wall_instr_per_simd_per_ns"] for r in rows if int(r["waves_per_simd"]) == 8}
    out = {"tag": a.tag, "device": dev, "rows": rows,
           # wall-clock ceilings at 8 waves/SIMD (the shader clock sags under a full-chip VALU load, so cycles x 2.4 GHz is not a rate): wave64 instructions per SIMD and ns
           "instr_per_simd_per_ns_fp32": rate.get("v_fma_f32"), "instr_per_simd_per_ns_int": rate.get("v_add_u32"),
           "instr_per_simd_per_ns_mixed": rate.get("v_cvt_f32_ubyte0 + v_fma_f32 (1:1)"), "instr_per_simd_per_ns_node_test": rate.get("v_pk_fma_f32 + 2 v_cvt_f32_ubyte0 (node test)"),
           "cycles_per_wave64_valu": sat.get("v_fma_f32"), "cycles_per_wave64_pk_fma": sat.get("v_pk_fma_f32"),
           "cycles_per_wave64_cvt_ubyte": sat.get("v_cvt_f32_ubyte0"), "cycles_per_wave64_mov": sat.get("v_mov_b32"),
           "source": f"profiles/{a.tag}_valu_issue_calibration.txt"}
    if a.pmc_db and os.path.exists(a.pmc_db):
        prow = pmc_rows(a.pmc_db)
        names = None
        if a.pmc_log:
            _, pl = parse(a.pmc_log)
            names = [(r["op"], int(r["waves_per_simd"])) for r in pl]
        timed = prow[1::2]  # every timed launch follows its warm-up launch
        lines.append("")
        lines.append("# rocprofv3 --pmc of `valu_calib --pmc` (timed launches): what the SQ counters report for a known instruction count")
        lines.append(f"{'instruction':38s} {'waves/SIMD':>10s} {'SQ_INSTS_VALU':>16s} {'expected':>16s} {'SQ_ACTIVE_INST_VALU':>20s} {'ACTIVE/INSTS':>13s} {'SQ_BUSY_CYCLES':>16s} {'SQ_WAVE_CYCLES':>16s}")
        pm = []
        for i, p in enumerate(timed):
            op, w = names[i] if names and i < len(names) else ("?", 0)
            waves = p.get("SQ_WAVES", 0.0)
            reps = 2000 // 4 if "dependent" in op else 2000
            expected = waves * reps * (96 if "node test" in op else 64)
            insts, act = p.get("SQ_INSTS_VALU", 0.0), p.get("SQ_ACTIVE_INST_VALU", 0.0)
            lines.append(f"{op:38s} {w:10d} {insts:16.0f} {expected:16.0f} {act:20.0f} {(act / insts if insts else 0):13.4f} {p.get('SQ_BUSY_CYCLES', 0.0):16.0f} {p.get('SQ_WAVE_CYCLES', 0.0):16.0f}")
            pm.append({"op": op, "waves_per_simd": w, **{k: v for k, v in p.items() if k != "kernel"}, "expected_valu_insts": expected})
        out["pmc"] = pm
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", f"{a.tag}_valu_issue_calibration.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_issue_calibration.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
