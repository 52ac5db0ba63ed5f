cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
run() { label=$1; shift; out=$(env "$@" 2>&1 | tail -1); echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r.get("stage_ms_per_step"), r.get("nodes_per_ray"), r.get("tris_per_ray"))' 2>/dev/null || echo "FAILED: $out" | cut -c1-300)"; }
for v in ${VARIANTS:-GATLING_DYN_FLUSH=0 GATLING_DYN_FLUSH=4 GATLING_DYN_FLUSH=8 GATLING_DYN_FLUSH=16 GATLING_DYN_FLUSH=32 GATLING_DYN_FLUSH=64}; do
  echo "== $v"
  for W in ${WORKLOADS:-c3 c4}; do run $W $v timeout 300 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-pmc; done
  run c5 $v timeout 400 python bench.py --workload c5 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc
done 2>&1 | tee $O/r02l_flush.txt
[ -n "$SKIP_TESTS" ] || { timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "baseline or traversal or cutout or soup or interior or instanc" 2>&1 | tail -3; }
