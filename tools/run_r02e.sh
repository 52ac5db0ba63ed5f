cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], r["stage_ms_per_step"]["traceMs"])'
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
echo "== c2 k_path (bw off)"; GATLING_PATH_BW=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
for PW in 96 128 160 192 256; do for T in 64 48 32; do
  echo "== c2 bw PW=$PW thr=$T"; GATLING_PATH_BW_PATHS=$PW GATLING_PATH_BW_SHADE=$T GATLING_PATH_BW_REGEN=$T timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
done; done
echo "== c1 bw default"; timeout 300 python bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | python -c "$J"
