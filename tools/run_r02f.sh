cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], {k: r.get(k) for k in ("valu_frac","valu_lane_utilisation","wait_inst_any_frac")}, {k: v.get("SQ_INSTS_VALU") for k, v in (r.get("pmc_kernels") or {}).items()})'
echo "== c2 bw default with PMC"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "$J"
for PW in 96 104; do for T in 40 48 56; do for D in 16 32 48; do
  echo "== c2 bw PW=$PW thr=$T dry=$D"; GATLING_PATH_BW_PATHS=$PW GATLING_PATH_BW_SHADE=$T GATLING_PATH_BW_REGEN=$T GATLING_PATH_BW_DRY=$D timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
done; done; done
echo "== split thresholds"; for S in 32 48 64; do for R in 16 32 64; do echo "shade=$S regen=$R"; GATLING_PATH_BW_SHADE=$S GATLING_PATH_BW_REGEN=$R timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"; done; done
