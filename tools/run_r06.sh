#!/bin/bash
# Round-6 GPU sessions, one script with selectable stages:
#   gpurun --timeout 1500 -- 'bash tools/run_r06.sh <tag> quick variants lowspp tests bench prof stallpmc'
# Everything lands under gpurun_out/<tag>_* (merged back by gpurun); copy what should be judged into profiles/ afterwards.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
TAG=$1; shift
J='import json,sys
j=json.loads(sys.stdin.read()); r=j["roofline"]
print(j["value"], j["ms_per_step"], r.get("stage_ms_per_step"), {k: r.get(k) for k in ("bound","frac","valu_frac","valu_lane_utilisation","l2_hit_rate","nodes_per_ray","tris_per_ray","pmc_note")})
for a in (j.get("also") or []):
    ar = a.get("roofline") or {}
    print("  also", a.get("workload"), a.get("value"), a.get("ms_per_step"), a.get("error"), a.get("stage_ms"), a.get("projected_8gpu"), {k: ar.get(k) for k in ("frac","valu_frac","valu_lane_utilisation","l2_hit_rate","nodes_per_ray","tris_per_ray","avg_launch_us","pmc_note")})
print("line bytes", len(json.dumps(j)))'
VARIANTS=${VARIANTS:-"default"}
for STAGE in "$@"; do
case $STAGE in
quick)  # the traversal-facing parity tests first: a broken kernel should cost one minute, not the session
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${QUICK_K:-traversal or golden or deep_trees or two_level or cutout or soup or instanced or interior or nee}" --durations=5 > $O/${TAG}_pytest_quick.log 2>&1; tail -12 $O/${TAG}_pytest_quick.log
  ;;
tests)
  GATLING_BUILD_TIMING=1 timeout 1800 python -m pytest tests -x -q -m gpu --durations=12 > $O/${TAG}_pytest_gpu.log 2>&1; tail -18 $O/${TAG}_pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  ;;
variants)  # A/B of prebuilt library variants (tools/build_variant.py) on the stage timers, per workload
  for V in $VARIANTS; do
    L=""; [ $V != default ] && L=$GRAFT_REPO_ROOT/gatling_amd/variants/libgatling_gi_$V.so
    for WS in ${VARIANT_WORKLOADS:-c3:32 c4:32 c5:8}; do
      W=${WS%%:*}; S=${WS##*:}
      echo "== $V $W spp $S" | tee -a $O/${TAG}_variants.txt
      GATLING_GI_LIB=$L timeout 600 python tools/gpu_variants.py $W $S ${VARIANT_ENVS:--} 2>&1 | grep -v "^\[gatling_gi\]" | tee -a $O/${TAG}_variants.txt
    done
  done
  ;;
slowtest)  # the whole C2 frame at spp 1024 against the oracle on all host cores (~3 min); log -> profiles/
  GATLING_SLOW_TESTS=1 timeout 900 python -m pytest tests/test_gpu_full_spp.py -x -q -s -m gpu -k whole_frame > $O/${TAG}_c2_whole_frame_spp1024.log 2>&1; tail -4 $O/${TAG}_c2_whole_frame_spp1024.log
  ;;
bigscene)  # 67.7 M flattened triangles -> automatic two-level layout (slow: ~10 GB of host arrays on both sides)
  GATLING_SLOW_TESTS=1 GATLING_BUILD_TIMING=1 timeout 1200 python -m pytest tests/test_gpu_full_spp.py -x -q -s -m gpu -k beyond_2_pow_26 > $O/${TAG}_scene_beyond_2p26.log 2>&1; tail -6 $O/${TAG}_scene_beyond_2p26.log
  ;;
lowspp)  # the delegate's default workload: one giRender per frame at spp 1 / 4 / 16, 13 bounces, progressive
  for V in ${LOWSPP_VARIANTS:-default}; do
    L=""; [ $V != default ] && L=$GRAFT_REPO_ROOT/gatling_amd/variants/libgatling_gi_$V.so
    echo "== $V" | tee -a $O/${TAG}_lowspp.txt
    GATLING_GI_LIB=$L timeout 900 python tools/lowspp.py ${LOWSPP_WORKLOADS:-c3,c4,c2} ${LOWSPP_SPPS:-1,4,16} 20 2>&1 | grep -v "^\[gatling_gi\]" | tee -a $O/${TAG}_lowspp.txt
  done
  ;;
lowsppprof)  # kernel trace of the spp-1 calls
  for W in ${LOWSPP_PROF_WORKLOADS:-c4}; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/${TAG}_lowspp_$W -o $W -- python $GRAFT_REPO_ROOT/tools/lowspp.py $W 1 20 > $O/${TAG}_prof_lowspp_$W.log 2>&1)
    DB=$(find $O/prof/${TAG}_lowspp_$W -name "*_results.db" | head -1)
    python tools/summarize_profile.py --kernel-trace $DB --tag ${TAG}_lowspp_$W --workload $W --spp 1 > $O/${TAG}_lowspp_summary_$W.txt 2>&1; head -24 $O/${TAG}_lowspp_summary_$W.txt | cut -c1-180
    cp profiles/${TAG}_lowspp_${W}_rocprofv3_summary.* $O/ 2>/dev/null
  done
  ;;
overlap)  # two-stream evidence: kernel trace of spp-1 calls with and without the second stream, time with two kernels running
  for M in 1 0; do
    (cd /tmp && GATLING_OPTIONS=two_stream=$M timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/${TAG}_ts$M -o c3 -- python $GRAFT_REPO_ROOT/tools/lowspp.py c3 1 20 > $O/${TAG}_prof_ts$M.log 2>&1)
    DB=$(find $O/prof/${TAG}_ts$M -name "*_results.db" | head -1)
    python tools/overlap_from_trace.py $DB ${TAG}_lowspp_c3_two_stream$M | tee $O/${TAG}_overlap_ts$M.txt
    python tools/summarize_profile.py --kernel-trace $DB --tag ${TAG}_lowspp_c3_two_stream$M --workload c3 --spp 1 > /dev/null 2>&1
    cp profiles/${TAG}_lowspp_c3_two_stream$M* $O/ 2>/dev/null
  done
  ;;
bench)
  timeout 2400 python bench.py ${BENCH_ARGS:-} > $O/${TAG}_bench_c2.log 2>&1; tail -1 $O/${TAG}_bench_c2.log | python -c "$J"
  cp $O/bench_last_pmc.json $O/${TAG}_bench_pmc.json 2>/dev/null
  ;;
benchquick)  # no PMC passes, no CPU baseline
  timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/${TAG}_benchq_c2.log 2>&1; tail -1 $O/${TAG}_benchq_c2.log | python -c "$J"
  ;;
prof)  # rocprofv3 --kernel-trace --stats of the bench command, per workload -> profiles/<tag>_<w>_rocprofv3_summary.*
  for WS in ${PROF_WORKLOADS:-c2:1024 c3:256 c4:256}; do W=${WS%%:*}; S=${WS##*:}
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/${TAG}_kt_$W -o $W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --spp $S --steps 1 --warmup 1 --no-timers --no-cpu-baseline --no-pmc > $O/${TAG}_prof_kt_$W.log 2>&1)
    DB=$(find $O/prof/${TAG}_kt_$W -name "*_results.db" | head -1)
    python tools/summarize_profile.py --kernel-trace $DB --tag ${TAG}_$W --workload $W --spp $S > $O/${TAG}_summary_$W.txt 2>&1; head -14 $O/${TAG}_summary_$W.txt | cut -c1-160
    cp profiles/${TAG}_${W}_rocprofv3_summary.* $O/ 2>/dev/null
  done
  ;;
stallpmc)  # where a kernel's wave-cycles go: issue, waits, instruction fetch (separate --pmc passes, no trace options)
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH_LEVEL" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
    N=$(echo $SET | tr ' ' '_' | cut -c1-40)
    (cd /tmp && timeout 300 rocprofv3 --pmc $SET -d $O/prof/${TAG}_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/gpu_variants.py ${STALL_WORKLOAD:-c3} ${STALL_SPP:-32} - > $O/${TAG}_pmc_$N.log 2>&1)
    DB=$(find $O/prof/${TAG}_$N -name "*_results.db" | head -1)
    [ -n "$DB" ] && python tools/pmc_dump.py $DB | grep "${STALL_KERNELS:-k_shade\|k_trace_dyn<false, false\|^#}" >> $O/${TAG}_stall_counters.txt
  done
  cat $O/${TAG}_stall_counters.txt
  ;;
valumix)
  for WS in ${MIX_WORKLOADS:-c3:32}; do W=${WS%%:*}; S=${WS##*:}
    echo "== $W spp $S" | tee -a $O/${TAG}_valu_mix.txt
    timeout 900 python tools/valu_mix.py $W $S 2>&1 | grep -v "^\[gatling_gi\]" | tee -a $O/${TAG}_valu_mix.txt
  done
  ;;
esac
done
find $O/prof -name "*.db" -size +6M -delete 2>/dev/null
find $O/prof -type f -size +6M -delete 2>/dev/null
du -sh $O | tail -1
