cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
run() { # label, env..., workload args
  label=$1; shift
  out=$(env "$@" 2>&1 | tail -1)
  echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null || echo "FAILED: $out" | cut -c1-300)"
}
for v in ${VARIANTS:-"GATLING_BVH_CPRIM=0.1" "GATLING_BVH_CPRIM=0.15" "GATLING_BVH_CPRIM=0.2" "GATLING_BVH_CPRIM=0.25"}; do
  echo "== $v"
  run c3 $v timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc
  run c2 $v timeout 300 python bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc
  run c4 $v timeout 300 python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc
  run c5 $v timeout 400 python bench.py --workload c5 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc
done 2>&1 | tee $O/r02j_collapse.txt
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traversal or bvh or c3 or two_level or cutout or golden" > $O/pytest_r02j.log 2>&1; tail -3 $O/pytest_r02j.log
