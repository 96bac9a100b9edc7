#!/bin/bash
# Round 4, call u: the bidirectional workloads - shade kernels launched with the resident workgroups (ETX_HIP_GRID_BDPT_SHADE, percent) and the light path table length.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4u
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['working_set_gb'])"; }
run() {  # workload name env...
  w=$1; name=$2; shift 2
  r=$(env "$@" ETX_HIP_LIBRARY=$V/libetx_hip_dbgb.so timeout 200 python bench.py --workload $w --steps 16 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
  echo "$w $name: $r" >> $O/ab.txt
}
run sssdragon_bdpt "grid uncapped, table 32" ETX_HIP_GRID_BDPT_SHADE=0
run sssdragon_bdpt "grid 100, table 32" ETX_HIP_GRID_BDPT_SHADE=100
run sssdragon_bdpt "grid uncapped, table 64" ETX_HIP_GRID_BDPT_SHADE=0 ETX_HIP_PATH_TABLE=64
run sssdragon_bdpt "grid 100, table 64" ETX_HIP_GRID_BDPT_SHADE=100 ETX_HIP_PATH_TABLE=64
run cloud_bdpt "grid uncapped" ETX_HIP_GRID_BDPT_SHADE=0
run cloud_bdpt "grid 100" ETX_HIP_GRID_BDPT_SHADE=100
run sssdragon_bdpt "grid uncapped, table 32 (again)" ETX_HIP_GRID_BDPT_SHADE=0
run cloud_bdpt "grid uncapped (again)" ETX_HIP_GRID_BDPT_SHADE=0
cat $O/ab.txt; tail -3 $O/err.txt
