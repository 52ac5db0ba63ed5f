set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 900 python tools/gpu_variants.py c3 16 GATLING_TRACE_DYN=0 GATLING_TRACE_DYN=16 GATLING_TRACE_DYN=32 GATLING_TRACE_DYN=48 GATLING_TRACE_DYN=8 GATLING_TRACE_DYN=32,GATLING_TRACE_DYN_SPILL8=1 GATLING_TRACE_DYN=32,GATLING_TRACE_BLOCKS_PER_CU=4 > gpurun_out/c3_variants.log 2>&1
tail -15 gpurun_out/c3_variants.log
timeout 600 python tools/gpu_variants.py c4 16 GATLING_TRACE_DYN=0 GATLING_TRACE_DYN=16 GATLING_TRACE_DYN=32 GATLING_TRACE_DYN=32,GATLING_TRACE_DYN_SPILL8=1 > gpurun_out/c4_variants.log 2>&1
tail -8 gpurun_out/c4_variants.log
cd /tmp
for v in 0 32; do
GATLING_TRACE_DYN=$v timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/gpurun_out/prof/sq$v -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --spp 4 --steps 1 --warmup 0 --no-timers --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_sq$v.log 2>&1
GATLING_TRACE_DYN=$v timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $GRAFT_REPO_ROOT/gpurun_out/prof/tcc$v -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --spp 4 --steps 1 --warmup 0 --no-timers --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_tcc$v.log 2>&1
GATLING_TRACE_DYN=$v timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum -d $GRAFT_REPO_ROOT/gpurun_out/prof/tcp$v -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --spp 4 --steps 1 --warmup 0 --no-timers --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_tcp$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_dump.py $(find gpurun_out/prof -name "*_results.db" | sort) > gpurun_out/c3_pmc.txt 2>&1
grep -E "k_trace|k_route" gpurun_out/c3_pmc.txt | head -80
find gpurun_out/prof -name "*.db" -size +20M -delete
