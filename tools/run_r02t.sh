# LDS-side counters of the fused kernel on C2 (is anything but VALU issue in the way?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload c2 --spp 64 --steps 1 --warmup 0 --no-timers --no-cpu-baseline --no-pmc"
i=0
for G in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_LDS SQ_WAVES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G -d $O/prof/lds_$i -o p -- $B > $O/prof_lds_$i.log 2>&1 || tail -3 $O/prof_lds_$i.log
done
cd $GRAFT_REPO_ROOT
python tools/pmc_dump.py $(find gpurun_out/prof -path "*lds_*" -name "*_results.db" | sort) 2>&1 | grep -E "^#|k_path_bw<1u" > gpurun_out/r02t_c2_lds_pmc.txt
cat gpurun_out/r02t_c2_lds_pmc.txt
