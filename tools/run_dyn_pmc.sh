# PMC characterisation of k_trace_dyn on C3 (one counter group per pass); summaries via tools/pmc_dump.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > gpurun_out/sq_counters.txt
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload c3 --spp 32 --steps 1 --warmup 0 --no-timers --no-cpu-baseline"
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G -d $GRAFT_REPO_ROOT/gpurun_out/prof/dyn$i -o c3 -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_dyn$i.log 2>&1 || tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_dyn$i.log
done
cd $GRAFT_REPO_ROOT
python tools/pmc_dump.py $(find gpurun_out/prof -path "*dyn*" -name "*_results.db" | sort) > gpurun_out/dyn_pmc.txt 2>&1
grep -E "k_trace_dyn<false, false" gpurun_out/dyn_pmc.txt
find gpurun_out/prof -name "*.db" -size +20M -delete
