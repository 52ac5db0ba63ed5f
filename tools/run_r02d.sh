cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], r["stage_ms_per_step"], {k: r.get(k) for k in ("bound","frac","valu_frac","valu_lane_utilisation","l2_hit_rate","traffic","traffic_upper","wait_inst_any_frac","pmc_note")})'
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log | python -c "$J"
for W in c1 c4 c3; do timeout 900 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$W.log 2>&1; echo "== $W"; tail -1 $O/bench_$W.log | python -c "$J"; done
timeout 900 python bench.py --workload c5 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.log 2>&1; echo "== c5@64"; tail -1 $O/bench_c5.log | python -c "$J"
