#!/bin/bash
# Round 4, call t: grids of the persistent kernels capped at the workgroups the device holds at once (kernels.h persistent_grid), per kernel family.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4t
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
run() {  # name lib env...
  name=$1; lib=$2; shift 2
  r=$(env "$@" ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
  echo "full $name: $r" >> $O/ab.txt
}
for round in 1 2; do
  run "shade 100, others uncapped" $V/libetx_hip_dbgk.so ETX_HIP_GRID_EXPAND=0 ETX_HIP_GRID_CONNECT=0 ETX_HIP_GRID_MERGE=0 ETX_HIP_GRID_SHADOW=0
  run "all 100" $V/libetx_hip_dbgk.so ETX_HIP_GRID_SHADE=100
  run "all 200" $V/libetx_hip_dbgk.so ETX_HIP_GRID_SHADE=200 ETX_HIP_GRID_EXPAND=200 ETX_HIP_GRID_CONNECT=200 ETX_HIP_GRID_MERGE=200 ETX_HIP_GRID_SHADOW=200
  run "all 50" $V/libetx_hip_dbgk.so ETX_HIP_GRID_SHADE=50 ETX_HIP_GRID_EXPAND=50 ETX_HIP_GRID_CONNECT=50 ETX_HIP_GRID_MERGE=50 ETX_HIP_GRID_SHADOW=50
  run "all 100, tables not staged" $V/libetx_hip_dbgk_nostage.so ETX_HIP_GRID_SHADE=100
  run "all 100, product build" etx-tracer_amd/libetx_hip.so ETX_HIP_GRID_SHADE=100
done
cat $O/ab.txt; tail -3 $O/err.txt
