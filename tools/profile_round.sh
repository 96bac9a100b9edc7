#!/bin/bash
# One profiling session on the GPU box (gpurun): kernel statistics of the default bench.py command, then PMC passes over a
# one-lane run (counters of concurrently running kernels of other lanes would mix). Outputs under gpurun_out/prof_<tag>/.
# Usage: tools/profile_round.sh <tag> [bench args...]
set -u
tag=$1; shift
bench_args=("$@")
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $root/bench.py --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline "$@" > $out/bench_stats.json 2> $out/bench_stats.err
export ETX_HIP_LANES=1
pass() {  # name counters...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out/pmc_$name -o pmc -- python $root/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-kernel-table "${bench_args[@]}" > $out/pmc_$name.log 2>&1
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU
pass tcc TCC_HIT_sum TCC_MISS_sum
pass derived OccupancyPercent VALUBusy MemUnitStalled
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $root
python3 tools/pmc_aggregate.py --meta $out/pmc_fetch.log $out/pmc_summary.json $(find $out -name "*counter_collection.csv" | sort) > $out/pmc_summary.txt 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
# the raw per-dispatch rows are large: keep the per-kernel aggregate, the kernel statistics and the two HBM passes
find $out -name "*kernel_trace.csv" -delete
for d in sq sq2 tcc derived; do rm -rf $out/pmc_$d; done
du -sh $out; ls $out
