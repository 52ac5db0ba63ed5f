cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], {k: r.get(k) for k in ("valu_frac","valu_lane_utilisation","wait_inst_any_frac")}, {k[:24]: (v.get("SQ_INSTS_VALU"), v.get("SQ_INSTS_LDS"), v.get("dispatches")) for k, v in (r.get("pmc_kernels") or {}).items()})'
echo "== c2 bw default with PMC"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "$J"
echo "== c2 k_path with PMC"; GATLING_PATH_BW=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "$J"
