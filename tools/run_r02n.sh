# PMC characterisation of EVERY kernel of a frame (C5 at spp 16, C3 at spp 32): HBM bytes, L2, VALU per kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/prof
cd /tmp
for W in "c5 16" "c3 32"; do set -- $W
  B="python $GRAFT_REPO_ROOT/bench.py --workload $1 --spp $2 --steps 1 --warmup 0 --no-timers --no-cpu-baseline --no-pmc"
  i=0
  for G in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $G -d $O/prof/all_$1_$i -o p -- $B > $O/prof_all_$1_$i.log 2>&1 || tail -3 $O/prof_all_$1_$i.log
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r02n_all_kernels_pmc.txt
import sqlite3, glob, collections
for w in ("c5", "c3"):
    agg = collections.defaultdict(dict)
    for path in sorted(glob.glob(f"gpurun_out/prof/all_{w}_*/**/*_results.db", recursive=True)):
        cur = sqlite3.connect(path).cursor()
        for k, c, n, v, dur in cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value), sum(duration) * 1.0 / count(*) from counters_collection group by kernel_name, counter_name"):
            k = k.replace("gi::", "").split("(")[0].replace("void ", "")
            agg[k][c] = v; agg[k]["n"] = n; agg[k].setdefault("us", dur / 1000.0)
    print(f"# {w}: per kernel: dispatches, mean us (under counters), FETCH_SIZE / WRITE_SIZE in bytes as reported (KiB x 1024; coalesced read streams are under-reported by 2, profiles/r02a_pmc_calibration.txt), L2 hit, VALU instr, lane util")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("us", 0) * kv[1].get("n", 0)):
        n, us = a.get("n", 0), a.get("us", 0.0)
        f, wr = a.get("FETCH_SIZE", 0) * 1024.0, a.get("WRITE_SIZE", 0) * 1024.0
        hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
        vi, tc = a.get("SQ_INSTS_VALU", 0), a.get("SQ_THREAD_CYCLES_VALU", 0)
        tot_s = n * us * 1e-6
        print(f"{k:46s} n {n:5d} us {us:10.1f} total_ms {n*us/1000:9.1f} fetchGB {f/1e9:9.2f} writeGB {wr/1e9:9.2f} raw_TBps {(f+wr)/max(tot_s,1e-12)/1e12:6.2f} l2hit {hit/max(hit+miss,1):5.2f} valuGinst {vi/1e9:8.2f} valu_frac {vi*4/(1024*2.4e9*max(tot_s,1e-12)):5.2f} lanes {tc/max(vi*64,1):5.2f}")
PY
cat gpurun_out/r02n_all_kernels_pmc.txt
find $O/prof -name "*.db" -size +8M -delete
