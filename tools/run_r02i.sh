cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/measure_shard_overhead.py
GATLING_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | cut -c1-200
