cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log | cut -c1-900
timeout 300 python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c1.log 2>&1; tail -1 $O/bench_c1.log | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/kt_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 1 --warmup 1 --no-timers --no-cpu-baseline --no-pmc > $O/prof_kt_c2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_profile.py --kernel-trace $O/prof/kt_c2/c2_results.db --tag r02h_c2 --workload c2 --spp 1024 > $O/summary_c2.txt 2>&1; cat profiles/r02h_c2_rocprofv3_summary.txt | head -8
cp profiles/r02h_* $O/
python -c "import __graft_entry__ as g; g.smoke()"
find $O/prof -name "*.db" -size +8M -delete
