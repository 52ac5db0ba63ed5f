# round 2, second GPU session: k_path occupancy sweep, C3 PMC roofline, 6-wave k_trace_dyn, ray-order potential, tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], r["stage_ms_per_step"], {k: r.get(k) for k in ("frac","valu_frac","l2_hit_rate","traffic","traffic_upper","wait_inst_any_frac","wave_cycles_not_valu_frac","pmc_note")})'
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for B in auto 3 4 5 6 7; do
  echo "== c2 k_path blocks/CU=$B"; E=""; [ $B != auto ] && E="GATLING_PATH_BLOCKS_PER_CU=$B"
  env $E timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
done
echo "== c1"; timeout 300 python bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
echo "== c3 full with PMC"; timeout 900 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3.log 2>&1; tail -1 $O/bench_c3.log | python -c "$J"
for W in 5 6; do
  echo "== c3 spp32 waves=$W"; GATLING_DYN_WAVES=$W timeout 300 python bench.py --workload c3 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
  echo "== c5 spp8 waves=$W"; GATLING_DYN_WAVES=$W timeout 600 python bench.py --workload c5 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
  echo "== c4 spp32 waves=$W"; GATLING_DYN_WAVES=$W timeout 600 python bench.py --workload c4 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
done
cd /tmp
for X in 0 1; do
  GATLING_DYN_XCD=$X timeout 600 rocprofv3 --kernel-trace -d $O/prof/order_$X -o ord -- python $GRAFT_REPO_ROOT/tools/exp_ray_order.py soup 1000000 > $O/order_$X.log 2>&1
  python - <<PY
import sqlite3
cur = sqlite3.connect("$O/prof/order_$X/ord_results.db").cursor()
rows = cur.execute("select name, duration from kernels where name like '%k_trace_dyn%' order by start").fetchall()
print("xcd=$X k_trace_dyn dispatch durations (us):", [round(r[1] / 1e3) for r in rows])
PY
  grep "order:" $O/order_$X.log | tr '\n' ';'; echo
done
find $O/prof -name "*.db" -size +8M -delete
