#!/bin/bash
# round 5, GPU call g: each lane on a quarter of the compute units of its own (hipExtStreamCreateWithCUMask; tools/experiments/round5_cu_partition.patch) against
# the shared device, interleaved; with the shade kernels' persistent grids at their full and at a quarter of their size.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5g
mkdir -p $O
export TMPDIR=/tmp
L=$PWD/etx-tracer_amd/variants/libetx_hip_cupart.so
for r in 1 2 3; do
  for combo in "0 100" "1 100" "1 25" "2 100" "2 25"; do
    set -- $combo
    x=$(ETX_HIP_LIBRARY=$L ETX_HIP_CU_PARTITION=$1 ETX_HIP_GRID_SHADE=$2 timeout 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
    echo "full partition=$1 shade-grid=$2% run $r: $x" >> $O/ab_cu_partition.txt
  done
done
for combo in "0 100" "1 100" "2 100"; do
  set -- $combo
  x=$(ETX_HIP_LIBRARY=$L ETX_HIP_CU_PARTITION=$1 timeout 200 python bench.py --workload classic --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
  echo "classic partition=$1: $x" >> $O/ab_cu_partition.txt
done
cat $O/ab_cu_partition.txt
