#!/bin/bash
# First-contact kit for a multi-GPU node (no round of this repo has had one): the headline bench at N = 1, 2, 4, 8 GPUs in BOTH multi-GPU forms --
#   ranks:      one process per GPU (torch.distributed.run, RCCL gather of the interleaved row shares over xGMI)
#   in-process: one process driving N devices inside the library (giCInitializeDevices; strided peer copies into place)
# -- one JSON line per (form, N) on stdout and in $OUT/scale_<form>_<N>.json, and at the end a table: Msamples/s, speed-up over N = 1, the frame's SHA-256
# (bench.py `image_checksum`: must be the SAME for every N and both forms -- rows are dealt to GPUs, the per-pixel arithmetic does not depend on the dealing),
# and what giCGetDevicePeerAccess says about every device (1 = peer copies, 0 / -1 = the library stages that device's shares through pinned host memory).
#   bash tools/run_scale.sh [max N, default: all GPUs of the node] [extra bench.py arguments, e.g. --steps 10]
# Exit code 1 if any checksum differs or a run fails.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=${TMPDIR:-/tmp} HSA_ENABLE_IPC_MODE_LEGACY=0 GATLING_BENCH_CHECKSUM=1 GATLING_BENCH_ALSO=
OUT=${OUT:-gpurun_out}; mkdir -p "$OUT"
HAVE=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
MAXN=${1:-$HAVE}; shift
[ "$HAVE" -ge 1 ] || { echo "no GPU visible: nothing to measure"; exit 1; }
echo "# $HAVE GPU(s) visible; measuring N <= $MAXN"
python - <<PY
import sys; sys.path.insert(0, ".")
from gatling_amd import capi
import torch
n = min(torch.cuda.device_count(), $MAXN)
L = capi.initialize(devices=list(range(n)))
print("# peer access of device i with device 0 (giCGetDevicePeerAccess):", {i: L.giCGetDevicePeerAccess(i) for i in range(n)})
PY
FAIL=0; PORT=29620
for N in 1 2 4 8; do
  [ "$N" -le "$MAXN" ] && [ "$N" -le "$HAVE" ] || continue
  for FORM in ranks in-process; do
    [ "$N" -eq 1 ] && [ "$FORM" = in-process ] && continue
    F="$OUT/scale_${FORM}_${N}.json"
    if [ "$FORM" = ranks ] && [ "$N" -gt 1 ]; then
      PORT=$((PORT + 1))
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$N" --no-pmc --no-cpu-baseline "$@" 2> "$F.err" | tail -1 > "$F"
    elif [ "$FORM" = in-process ]; then
      python bench.py --gpus "$N" --in-process --no-pmc --no-cpu-baseline "$@" 2> "$F.err" | tail -1 > "$F"
    else
      python bench.py --gpus 1 --no-pmc --no-cpu-baseline "$@" 2> "$F.err" | tail -1 > "$F"
    fi
    python -c "import json,sys; j=json.load(open('$F')); print(json.dumps({'form':'$FORM','n_gpus':$N,'value':j['value'],'ms_per_step':j['ms_per_step'],'image_checksum':j.get('image_checksum')}))" || { echo "FAILED: $FORM N=$N (see $F.err)"; FAIL=1; }
  done
done
python - "$OUT" <<'PY' || FAIL=1
import glob, json, os, sys
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "scale_*_*.json"))):
    try:
        j = json.load(open(f)); form, n = os.path.basename(f)[6:-5].rsplit("_", 1)
        rows.append((form, int(n), j["value"], j["ms_per_step"], j.get("image_checksum")))
    except Exception as e:  # noqa: BLE001
        print("unreadable:", f, e)
if not rows:
    sys.exit("no results")
base = next((r[2] for r in rows if r[1] == 1), rows[0][2])
print(f"{'form':12s} {'N':>2s} {'Msamples/s':>12s} {'ms/step':>9s} {'x N=1':>6s}  image sha256")
for form, n, v, ms, cs in sorted(rows, key=lambda r: (r[0], r[1])):
    print(f"{form:12s} {n:2d} {v:12.1f} {ms:9.2f} {v / base:6.2f}  {str(cs)[:16]}")
sums = {r[4] for r in rows}
if len(sums) != 1 or None in sums:
    sys.exit(f"image checksums differ across runs: {sums}")
print("image checksums equal across every N and both forms")
PY
exit $FAIL
