#!/bin/bash
# Round 4, call d: what bounds the tree traversal kernel (it did not react to occupancy, staged nodes, refill threshold or the tree's width in call c):
# the gather microbenchmark (lane-loads per clock and CU through the vector L1 / LDS at per-lane addresses) and PMC passes over the kernel alone.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$(pwd)/gpurun_out/r4d
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/micro/bin/gather_bench > $O/gather_bench.txt 2>&1
R=$(pwd)
cd /tmp
S=$R/tests/golden/cornell_gems_1080p.etxscene
n=0
for counters in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
                "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD" \
                "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" \
                "TA_TA_BUSY_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
                "OccupancyPercent VALUBusy MemUnitBusy MemUnitStalled"; do
  n=$((n+1))
  for tree in host wide; do
    t=""; [ $tree = wide ] && t=wide
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $counters -d $O/pmc_${tree}_$n -o pmc -- python $R/tools/trace_bench.py $S 2073600 5 $t > $O/pmc_${tree}_$n.log 2>&1 || echo "pass $n $tree failed" >> $O/log.txt
  done
done
cd $R
for tree in host wide; do
  python3 tools/pmc_aggregate.py $O/pmc_${tree}_summary.json $(find $O -path "*pmc_${tree}_*" -name "*counter_collection.csv" | sort) x x x x x x x > $O/pmc_${tree}_summary.txt 2>&1
done
rm -rf $O/pmc_host_? $O/pmc_wide_?
cat $O/gather_bench.txt
grep "k_trace_closest_bvh" -A0 $O/pmc_host_summary.txt | head -3; grep "k_trace_closest_bvh" $O/pmc_wide_summary.txt | head -3
cat $O/log.txt 2>/dev/null
