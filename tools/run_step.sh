cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --workload c5 --spp 16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c5.log 2>&1; tail -1 gpurun_out/bench_c5.log
timeout 900 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/bench_c4.log 2>&1; tail -1 gpurun_out/bench_c4.log
timeout 900 python bench.py --workload c3 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; tail -1 gpurun_out/bench_c3.log
