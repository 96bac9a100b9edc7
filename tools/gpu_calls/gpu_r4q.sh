#!/bin/bash
# Round 4, call q: LDS tables read through generic pointers (flat_load) in the gather microbenchmark; path table 32 (now the default) against 64 and 16.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4q
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['working_set_gb'])"; }
timeout 60 tools/micro/bin/gather_bench > $O/gather_bench.txt 2>&1
for round in 1 2; do
  for t in 16 32 64; do
    r=$(ETX_HIP_PATH_TABLE=$t ETX_HIP_LIBRARY=$V/libetx_hip_dbgapi.so timeout 120 python bench.py --workload full --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full path table $t: 4 lanes $r" >> $O/ab.txt
  done
done
cat $O/ab.txt; grep "LDS\|no load" $O/gather_bench.txt
