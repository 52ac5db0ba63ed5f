cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
GATLING_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1; tail -2 gpurun_out/bench_dist1.log | cut -c1-700
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --workload c5 --spp 64 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c5.log 2>&1; tail -1 gpurun_out/bench_c5.log | cut -c1-1500
