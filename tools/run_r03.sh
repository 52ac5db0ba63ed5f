#!/bin/bash
# Round-3 GPU sessions, one script with selectable stages (replaces the per-experiment run_r02*.sh one-shots):
#   gpurun --timeout 1500 -- 'bash tools/run_r03.sh <tag> calib tests variants bench prof order'
# Everything lands under gpurun_out/<tag>_* (merged back by gpurun); copy what should be judged into profiles/ afterwards.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
TAG=$1; shift
J='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], j["ms_per_step"], r["stage_ms_per_step"], {k: r.get(k) for k in ("bound","frac","valu_frac","valu_lane_utilisation","l2_hit_rate","traffic","nodes_per_ray","tris_per_ray","pmc_note")}); [print("  also", a.get("workload"), a.get("value"), a.get("ms_per_step"), a.get("error"), (a.get("roofline") or {}).get("stage_ms_per_step"), {k: (a.get("roofline") or {}).get(k) for k in ("frac","valu_frac","valu_lane_utilisation","l2_hit_rate")}) for a in (j.get("also") if isinstance(j.get("also"), list) else [j["also"]] if j.get("also") else [])]'
VARIANTS=${VARIANTS:-"default noslp1 noslp0 slp0"}
for STAGE in "$@"; do
case $STAGE in
calib)  # VALU issue-rate calibration (tools/valu_calib.hip) + what the SQ counters report for it
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_calib tools/valu_calib.hip 2>/dev/null
  timeout 300 /tmp/valu_calib > $O/${TAG}_valu_calib.log 2>&1
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -d $O/prof/${TAG}_valu -o valu -- /tmp/valu_calib --pmc > $O/${TAG}_valu_calib_pmc.log 2>&1)
  DB=$(find $O/prof/${TAG}_valu -name "*_results.db" | head -1)
  python tools/valu_calib_report.py $O/${TAG}_valu_calib.log $DB --pmc-log $O/${TAG}_valu_calib_pmc.log --tag $TAG > $O/${TAG}_valu_calib_report.txt 2>&1
  cp profiles/${TAG}_valu_issue_calibration.txt profiles/valu_issue_calibration.json $O/ 2>/dev/null
  cat $O/${TAG}_valu_calib_report.txt | cut -c1-200
  ;;
ta)  # vector-memory request-path calibration (tools/ta_calib.hip): per-lane record fetches per ns and CU, by pattern and working set
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_calib tools/ta_calib.hip 2>/dev/null
  timeout 600 /tmp/ta_calib > $O/${TAG}_ta_calib.txt 2>&1; cut -c1-230 $O/${TAG}_ta_calib.txt
  ;;
tests)
  GATLING_BUILD_TIMING=1 timeout 1500 python -m pytest tests -x -q -m gpu > $O/${TAG}_pytest_gpu.log 2>&1; tail -5 $O/${TAG}_pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  ;;
variants)  # A/B of prebuilt library variants (tools/build_variant.py) on the stage timers, per workload
  for V in $VARIANTS; do
    L=""; [ $V != default ] && L=$GRAFT_REPO_ROOT/gatling_amd/variants/libgatling_gi_$V.so
    for WS in ${VARIANT_WORKLOADS:-c3:32 c4:32 c5:8 c2:128}; do
      W=${WS%%:*}; S=${WS##*:}
      echo "== $V $W spp $S" | tee -a $O/${TAG}_variants.txt
      GATLING_GI_LIB=$L timeout 600 python tools/gpu_variants.py $W $S ${VARIANT_ENVS:--} 2>&1 | grep -v "^\[gatling_gi\]" | tee -a $O/${TAG}_variants.txt
    done
  done
  ;;
nodelines)  # one BVH8 node per 128-byte line (GATLING_NODE_LINES=1, read at scene build) vs the packed 80-byte stride
  for NL in 0 1; do for WS in c3:32 c4:32 c5:8; do W=${WS%%:*}; S=${WS##*:}
    echo "== node_lines=$NL $W spp $S" | tee -a $O/${TAG}_nodelines.txt
    GATLING_NODE_LINES=$NL timeout 600 python tools/gpu_variants.py $W $S - - 2>&1 | grep -v "^\[gatling_gi\]" | tee -a $O/${TAG}_nodelines.txt
  done; done
  ;;
newtests)
  timeout 900 python -m pytest tests/test_mtlx_parity.py tests/test_multi_device.py -x -q -m gpu 2>&1 | tail -5
  ;;
inprocess)  # bench.py --in-process: two device contexts on the one GPU of this box (the multi-device path inside the library), C2 at spp 128
  GATLING_BENCH_ALSO= timeout 600 python bench.py --spp 128 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
  GATLING_BENCH_ALSO= GATLING_DEVICES=0,0 timeout 600 python bench.py --spp 128 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
  ;;
twolevel)
  for WS in "c4:32" "c5:8"; do W=${WS%%:*}; S=${WS##*:}
    echo "== two-level $W spp $S" | tee -a $O/${TAG}_twolevel.txt
    GATLING_TWO_LEVEL=1 timeout 600 python tools/gpu_variants.py $W $S - 2>&1 | grep -v "^\[gatling_gi\]" | tee -a $O/${TAG}_twolevel.txt
  done
  ;;
bench)
  timeout 1200 python bench.py > $O/${TAG}_bench_c2.log 2>&1; tail -1 $O/${TAG}_bench_c2.log | python -c "$J"
  ;;
benchall)
  for W in c1 c5; do SP=""; [ $W = c5 ] && SP="--spp 64"
    timeout 900 python bench.py --workload $W $SP --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_$W.log 2>&1; echo "== $W"; tail -1 $O/${TAG}_bench_$W.log | python -c "$J"; done
  ;;
prof)  # rocprofv3 --kernel-trace --stats of the bench command, per workload -> profiles/<tag>_<w>_rocprofv3_summary.*
  for WS in ${PROF_WORKLOADS:-c2:1024 c3:256 c4:256}; do W=${WS%%:*}; S=${WS##*:}
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/${TAG}_kt_$W -o $W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --spp $S --steps 1 --warmup 1 --no-timers --no-cpu-baseline --no-pmc > $O/${TAG}_prof_kt_$W.log 2>&1)
    DB=$(find $O/prof/${TAG}_kt_$W -name "*_results.db" | head -1)
    python tools/summarize_profile.py --kernel-trace $DB --tag ${TAG}_$W --workload $W --spp $S > $O/${TAG}_summary_$W.txt 2>&1; head -14 $O/${TAG}_summary_$W.txt | cut -c1-160
    cp profiles/${TAG}_${W}_rocprofv3_summary.* $O/ 2>/dev/null
  done
  ;;
order)  # ray order vs traversal time where the BVH misses the caches
  for M in ${ORDER_MODES:-instances interior}; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $O/prof/${TAG}_order_kt_$M -o t -- python $GRAFT_REPO_ROOT/tools/exp_ray_order.py $M > $O/${TAG}_order_$M.log 2>&1)
    (cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/prof/${TAG}_order_pmc_$M -o t -- python $GRAFT_REPO_ROOT/tools/exp_ray_order.py $M > $O/${TAG}_order_pmc_$M.log 2>&1)
    KT=$(find $O/prof/${TAG}_order_kt_$M -name "*_results.db" | head -1); PM=$(find $O/prof/${TAG}_order_pmc_$M -name "*_results.db" | head -1)
    python tools/exp_ray_order_report.py $O/${TAG}_order_$M.log $KT $PM --out $O/${TAG}_ray_order_$M.txt 2>&1 | cut -c1-150
  done
  ;;
esac
done
find $O/prof -name "*.db" -size +6M -delete 2>/dev/null
find $O/prof -type f -size +6M -delete 2>/dev/null
du -sh $O | tail -1
