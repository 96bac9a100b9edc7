#!/bin/bash
# Round 4, call x: the tail threshold (active paths <= capacity / divisor: the pass ends in the tail kernel) with this round's faster shade kernels.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4x
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for round in 1 2; do
  for t in 16 32 64 128; do
    r=$(ETX_HIP_TAIL_DIVISOR=$t ETX_HIP_LIBRARY=$V/libetx_hip_dbgapi.so timeout 120 python bench.py --workload full --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    c=$(ETX_HIP_TAIL_DIVISOR=$t ETX_HIP_LIBRARY=$V/libetx_hip_dbgapi.so timeout 120 python bench.py --workload classic --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "tail divisor $t: full $r classic $c" >> $O/ab.txt
  done
done
cat $O/ab.txt
