# Round-end measurement set: GPU tests, default bench, C1/C3/C4 lines, rocprofv3 kernel trace + PMC traffic for C2 and C3.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01f}
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_c2.log 2>&1; tail -1 gpurun_out/bench_c2.log
timeout 600 python bench.py --workload c1 --no-cpu-baseline > gpurun_out/bench_c1.log 2>&1; tail -1 gpurun_out/bench_c1.log
timeout 600 python bench.py --workload c3 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; tail -1 gpurun_out/bench_c3.log
timeout 600 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/bench_c4.log 2>&1; tail -1 gpurun_out/bench_c4.log
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
cd /tmp
for W in c2 c3; do
  # full spp of the workload: per-launch PMC bytes are then directly comparable with bench.py's per-launch algorithmic bytes
  B="python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 1 --warmup 1 --no-timers --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt_$W -o $W -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_kt_$W.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof/fetch_$W -o $W -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_fetch_$W.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof/write_$W -o $W -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_write_$W.log 2>&1
done
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_new; 
for W in c2 c3; do
  SPP=1024; [ $W = c3 ] && SPP=256
  python tools/summarize_profile.py --kernel-trace gpurun_out/prof/kt_$W/${W}_results.db --fetch gpurun_out/prof/fetch_$W/${W}_results.db --write gpurun_out/prof/write_$W/${W}_results.db --tag ${TAG}_$W --workload $W --spp $SPP > gpurun_out/summary_$W.txt 2>&1
done
cp profiles/${TAG}_* profiles/pmc_traffic.json gpurun_out/profiles_new/ 2>/dev/null
cat gpurun_out/summary_c3.txt | head -30
find gpurun_out/prof -name "*.db" -size +20M -delete
