# round 2, first GPU session: regression tests, C2 bench with live PMC, kernel trace, PMC calibration, k_trace_dyn knob sweeps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log | cut -c1-1500
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 1 --warmup 1 --no-timers --no-cpu-baseline --no-pmc"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/kt_c2 -o c2 -- $B > $O/prof_kt_c2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof/calib_f -o calib -- $GRAFT_REPO_ROOT/tools/build/pmc_calib > $O/calib_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof/calib_w -o calib -- $GRAFT_REPO_ROOT/tools/build/pmc_calib > $O/calib_w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_calib_report.py $O/prof/calib_f/calib_results.db $O/prof/calib_w/calib_results.db --tag r02a > $O/calib_report.txt 2>&1; cat $O/calib_report.txt
python tools/summarize_profile.py --kernel-trace $O/prof/kt_c2/c2_results.db --tag r02a_c2 --workload c2 --spp 1024 > $O/summary_c2.txt 2>&1; cat profiles/r02a_c2_rocprofv3_summary.txt | head -12
for L in 0 9 73 585; do for X in 0 1; do
  echo "== c3 spp32 lds=$L xcd=$X"; GATLING_DYN_LDS_NODES=$L GATLING_DYN_XCD=$X timeout 300 python bench.py --workload c3 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['stage_ms_per_step'])"
done; done
for L in 0 73; do
  echo "== c5 spp8 lds=$L"; GATLING_DYN_LDS_NODES=$L timeout 600 python bench.py --workload c5 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['stage_ms_per_step'])"
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
cd $GRAFT_REPO_ROOT
cp profiles/r02a_* profiles/pmc_calibration.json $O/ 2>/dev/null
find $O/prof -name "*.db" -size +8M -delete
