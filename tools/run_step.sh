cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py --workload c3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 900 python bench.py --workload c4 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 900 python bench.py --workload c5 --spp 64 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
