cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload c3 --spp 4 --steps 1 --warmup 0 --no-timers --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/gpurun_out/prof/sq1 -o c3 -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU -d $GRAFT_REPO_ROOT/gpurun_out/prof/sq2 -o c3 -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_sq2.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $GRAFT_REPO_ROOT/gpurun_out/prof/tcp -o c3 -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_tcp.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $GRAFT_REPO_ROOT/gpurun_out/prof/tcc -o c3 -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_tcc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt -o c3 -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_dump.py $(find gpurun_out/prof -name "*_results.db" | sort) > gpurun_out/c3_pmc.txt 2>&1
grep -E "k_trace_dyn<false, false" gpurun_out/c3_pmc.txt
tail -3 gpurun_out/prof_sq2.log
python - <<'PY'
import sqlite3
cur=sqlite3.connect('gpurun_out/prof/kt/c3_results.db').cursor()
for r in cur.execute("select name, count(*), sum(duration)/1e6, avg(duration)/1e3 from kernels group by name order by sum(duration) desc").fetchall(): print(r[0][:70], r[1], round(r[2],3), round(r[3],1))
PY
find gpurun_out/prof -name "*.db" -size +20M -delete
