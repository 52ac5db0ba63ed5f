cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], r["stage_ms_per_step"], {k: r.get(k) for k in ("bound","frac","valu_frac","valu_lane_utilisation","l2_hit_rate","traffic","nodes_per_ray","tris_per_ray","pmc_note")})'
GATLING_BUILD_TIMING=1 timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log | python -c "$J"
for W in c1 c4 c3; do timeout 900 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$W.log 2>&1; echo "== $W"; tail -1 $O/bench_$W.log | python -c "$J"; done
timeout 900 python bench.py --workload c5 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.log 2>&1; echo "== c5@64"; tail -1 $O/bench_c5.log | python -c "$J"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/kt_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 1 --warmup 1 --no-timers --no-cpu-baseline --no-pmc > $O/prof_kt_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/kt_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 1 --warmup 1 --no-timers --no-cpu-baseline --no-pmc > $O/prof_kt_c3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_profile.py --kernel-trace $O/prof/kt_c2/c2_results.db --tag r02s_c2 --workload c2 --spp 1024 > $O/summary_c2.txt 2>&1; head -12 profiles/r02s_c2_rocprofv3_summary.txt
python tools/summarize_profile.py --kernel-trace $O/prof/kt_c3/c3_results.db --tag r02s_c3 --workload c3 --spp 256 > $O/summary_c3.txt 2>&1; head -12 profiles/r02s_c3_rocprofv3_summary.txt
cp profiles/r02s_* $O/
python -c "import __graft_entry__ as g; g.smoke()"
find $O/prof -name "*.db" -size +8M -delete
