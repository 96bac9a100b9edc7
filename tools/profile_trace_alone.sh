#!/bin/bash
# The tree traversal kernel ALONE on the device under rocprofv3 counters ("find out what bounds the tree kernels").
# tools/trace_bench.py traces 2 M camera-like and 2 M incoherent rays, `repeat` launches each, on device-resident queues; here under
# separate --pmc passes (with --kernel-trace only, as the guide prescribes; no TA_* counters: those crashed rocprofv3 in round 4).
#   pass sq     where the wave cycles go: waiting (s_waitcnt) / issue-stalled / issuing, VALU and VMEM issue shares
#   pass lanes  SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) = the share of LANES a VALU instruction has active (divergence between node
#               and leaf steps, finished rays), instruction counts per ray
#   pass tcc    L2 hit rate
# Usage: tools/profile_trace_alone.sh <tag> <snapshot>      -> gpurun_out/trace_alone_<tag>/summary.{json,txt}
set -u
tag=$1; snapshot=$2
root=$(pwd)
out=$root/gpurun_out/trace_alone_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out/pmc_$name -o pmc -- python $root/tools/trace_bench.py $snapshot 2073600 6 > $out/pmc_$name.log 2>&1 || echo "pass $name failed"
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum
cd $root
python3 tools/pmc_aggregate.py $out/summary.json $(find $out -name "*counter_collection.csv" | sort) > $out/summary.txt 2>&1
grep -h "Grays" $out/pmc_sq.log > $out/rates_under_counters.txt
find $out -name "*kernel_trace.csv" -delete
for d in sq lanes tcc; do rm -rf $out/pmc_$d; done
cat $out/summary.txt | head -40
