#!/bin/bash
# Round 4, call e: is the texture-addresser (TA) the unit the gather-heavy kernels wait for? Gather microbenchmark with LDS dword reads and a
# no-load control; TA / SQ counters of the tree traversal kernel alone; TA busy share per kernel of a one-lane configs[1] run.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
O=$R/gpurun_out/r4e
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/micro/bin/gather_bench > $O/gather_bench.txt 2>&1
cd /tmp
S=$R/tests/golden/cornell_gems_1080p.etxscene
n=0
for counters in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
                "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
                "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
                "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
                "OccupancyPercent VALUBusy MemUnitBusy MemUnitStalled"; do
  n=$((n+1))
  for tree in host wide; do
    t=""; [ $tree = wide ] && t=wide
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $counters -d $O/pmc_${tree}_$n -o pmc -- python $R/tools/trace_bench.py $S 2073600 5 $t > $O/pmc_${tree}_$n.log 2>&1 || echo "trace pass $n $tree failed" >> $O/log.txt
  done
done
export ETX_HIP_LANES=1
n=0
for counters in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
                "OccupancyPercent VALUBusy MemUnitBusy MemUnitStalled" \
                "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_FLAT"; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $counters -d $O/pmc_full_$n -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-table > $O/pmc_full_$n.log 2>&1 || echo "full pass $n failed" >> $O/log.txt
done
cd $R
for tag in host wide full; do
  python3 tools/pmc_aggregate.py $O/pmc_${tag}_summary.json $(find $O -path "*pmc_${tag}_*" -name "*counter_collection.csv" | sort) x x x x x x x > $O/pmc_${tag}_summary.txt 2>&1
done
rm -rf $O/pmc_host_? $O/pmc_wide_? $O/pmc_full_?/
cat $O/gather_bench.txt | tail -12
grep "k_trace_closest_bvh" $O/pmc_host_summary.txt | head -2; grep "k_trace_closest_bvh" $O/pmc_wide_summary.txt | head -2
head -8 $O/pmc_full_summary.txt
cat $O/log.txt 2>/dev/null
