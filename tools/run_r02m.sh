cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
run() { label="$1"; shift; out=$(env "$@" timeout 300 python bench.py --workload ${W:-c2} --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1); echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null || echo "FAILED: $out" | cut -c1-300)"; }
{
run "default" A=1
for s in 40 56 64; do run "shade=$s" GATLING_PATH_BW_SHADE=$s; done
for r in 16 24 40 48; do run "regen=$r" GATLING_PATH_BW_REGEN=$r; done
for d in 8 24 32; do run "dry=$d" GATLING_PATH_BW_DRY=$d; done
for p in 80 112 128; do run "paths=$p" GATLING_PATH_BW_PATHS=$p; done
run "shade=56 regen=40" GATLING_PATH_BW_SHADE=56 GATLING_PATH_BW_REGEN=40
run "shade=40 regen=24" GATLING_PATH_BW_SHADE=40 GATLING_PATH_BW_REGEN=24
run "k_path (no bw)" GATLING_PATH_BW=0
run "default" A=1
W=c1 run "c1 default" A=1
W=c1 run "c1 k_path" GATLING_PATH_BW=0
} 2>&1 | tee $O/r02m_bw_sweep.txt
