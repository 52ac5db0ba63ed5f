cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], r["stage_ms_per_step"])'
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
for W in c2 c4 c3; do echo "== $W"; timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"; done
