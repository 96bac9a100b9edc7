#!/bin/bash
# Ad-hoc PMC passes over a one-lane bench run: tools/profile_pmc.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...
# -> gpurun_out/pmc_<tag>/summary.{json,txt} (tools/pmc_aggregate.py); the raw rows are deleted.
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export ETX_HIP_LANES=1
n=0
for counters in "$@"; do
  n=$((n+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $counters -d $out/pass$n -o pmc -- python $root/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-kernel-table > $out/pass$n.log 2>&1 || echo "pass $n failed"
  grep -i "error\|invalid\|unable" $out/pass$n.log | head -3
done
cd $root
python3 tools/pmc_aggregate.py $out/summary.json $(find $out -name "*counter_collection.csv" | sort) > $out/summary.txt 2>&1
rm -rf $out/pass*/
