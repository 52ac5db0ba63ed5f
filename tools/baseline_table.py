#!/usr/bin/env python3
"""BASELINE.md section 4's result table from a bench record: the driver's BENCH_rNN.json (its `parsed` / last stdout line) or a file holding bench.py's JSON line.

  python tools/baseline_table.py BENCH_r05.json [more records ...]      -> markdown rows, one block per record

B_sample follows BASELINE.md section 4: per path segment 192 B of stream traffic + 220 B geometry fetch + 80 B per node visited + 48 B per triangle tested, plus 16 / spp B for
the colour write (NEE configurations: + 96 B per shadow ray, whose node / triangle bytes the line does not carry -- the figure is a lower bound there and says so)."""
import json
import sys

SAMPLES = {"c1": 512 * 512 * 64, "c2": 1920 * 1080 * 1024, "c3": 1920 * 1080 * 256, "c4": 1920 * 1080 * 256, "c5share": 270 * 3840 * 1024, "c5@64": 3840 * 2160 * 64}
NAMES = {"c1": "C1 cornell 512², 64 spp, 4 b, diffuse", "c2": "C2 cornell 1080p, 1024 spp, 8 b", "c3": "C3 1 M-tri, 1 OpenPBR, NEE, 1080p, 256 spp",
         "c4": "C4 32 BSDFs on instanced spheres, 1080p, 256 spp", "c5share": "C5 4K interior, 1024 spp: ONE rank's share of the 8-GPU partition (rows 3::8)", "c5@64": "C5 whole 4K frame at spp 64"}


def load(path):
    text = open(path).read()
    try:
        j = json.loads(text)
    except json.JSONDecodeError:
        j = json.loads(text.strip().splitlines()[-1])
    if "run" in j:  # the driver's record: the full line is the last line of the run's stdout
        try:
            line = [l for l in j["run"]["stdout_tail"].splitlines() if l.startswith('{"metric"')][-1]  # (the tail also holds the run's stderr)
            return json.loads(line), j.get("head", "?")
        except Exception:  # noqa: BLE001
            pass
    if "parsed" in j and isinstance(j["parsed"], dict) and "value" in j["parsed"]:
        return j["parsed"], j.get("head", "?")
    return j, "?"


def row(key, value, ms, seg, roof, cpu, nee):
    n = SAMPLES[key]
    npr, tpr = roof.get("nodes_per_ray"), roof.get("tris_per_ray")
    b = None
    if seg and npr is not None:
        b = seg * (192.0 + 220.0 + 80.0 * npr + 48.0 * tpr)
    frac = (b * value * 1e6 / 8e12) if b else None
    stream = seg * 192.0 * value * 1e6 / 8e12 if seg else None
    hbm = roof.get("achieved")
    cells = [NAMES[key], "1", f"{n / 1e6:.1f} M" if n < 1e9 else f"{n / 1e9:.3f} G", f"{ms / 1e3:.4f}", f"{value:,.0f}".replace(",", " "), f"{seg:.3f}" if seg else "",
             (f"{b:,.0f}".replace(",", " ") + (" (+ shadow rays)" if nee else "")) if b else "", f"{frac:.3f}" if frac else "", f"{stream:.3f}" if stream else "",
             f"{hbm:,.0f} ({roof.get('frac'):.3f} of 8 TB/s)".replace(",", " ") if hbm else "not collected for this leg", cpu, "bit-identical (tests/test_gpu_full_spp.py, tests/test_gpu_parity.py)"]
    return "| " + " | ".join(cells) + " |"


def main():
    for path in sys.argv[1:]:
        j, head = load(path)
        print(f"<!-- {path} (head {head}): python bench.py --steps {j.get('steps')} --warmup {j.get('warmup')} -->")
        cpu = j.get("cpu_baseline", {})
        cpu_s = f"{cpu.get('value')} ({cpu.get('cores')} cores, {cpu.get('kind')})" if cpu else ""
        print(row("c2", j["value"], j["ms_per_step"], j["config"].get("segments_per_sample"), j["roofline"], cpu_s, False))
        for e in j.get("also", []):
            k = e.get("workload")
            if k in SAMPLES and "value" in e:
                print(row(k, e["value"], e["ms_per_step"], e.get("segments_per_sample"), e.get("roofline", {}), "(C2's row: the oracle's rate is per sample)", k in ("c3", "c5share", "c5@64")))
        for e in j.get("also", []):
            if str(e.get("workload", "")).endswith("@spp1"):
                print(f"<!-- {e['workload']}: {e.get('ms_per_call')} ms per giCRender call (spp 1, 13 bounces, D2H included), {e.get('iterations_per_call')} iterations -->")


if __name__ == "__main__":
    main()
