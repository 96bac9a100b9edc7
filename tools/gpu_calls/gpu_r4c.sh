#!/bin/bash
# Round 4, call c: sweep of the tree traversal kernel's launch shape on both trees (debug variants of kernels_trace.hip read the ETX_HIP_*
# knobs; call b ran them against variants built before the ABI change), lanes vs throughput / memory of configs[1].
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4c
mkdir -p $O
export TMPDIR=/tmp
S=tests/golden/cornell_gems_1080p.etxscene
run() { # label env...
  label=$1; shift
  echo "== $label" >> $O/sweep.txt
  env "$@" timeout 120 python tools/trace_bench.py $S 2073600 20 $TREE 2>>$O/sweep.err | grep -v amdgpu.ids >> $O/sweep.txt
}
V=etx-tracer_amd/variants
for b in 1024 1536 2048; do
  TREE="" run "host blocks $b" ETX_HIP_LIBRARY=$V/libetx_hip_dbg.so ETX_HIP_DEBUG_BLOCKS=$b
  TREE=wide run "wide blocks $b" ETX_HIP_LIBRARY=$V/libetx_hip_dbg.so ETX_HIP_DEBUG_BLOCKS=$b
done
for b in 1536 2048; do
  TREE="" run "host short stack blocks $b" ETX_HIP_LIBRARY=$V/libetx_hip_dbg.so ETX_HIP_DEBUG_BLOCKS=$b ETX_HIP_BVH_VARIANT=2
done
for b in 768 1024; do
  TREE=wide run "wide 256 staged nodes blocks $b" ETX_HIP_LIBRARY=$V/libetx_hip_dbg256.so ETX_HIP_DEBUG_BLOCKS=$b ETX_HIP_LDS_NODES=256
done
for r in 8 32 48; do
  TREE=wide run "wide blocks 1536 refill $r" ETX_HIP_LIBRARY=$V/libetx_hip_dbg.so ETX_HIP_DEBUG_BLOCKS=1536 ETX_HIP_REFILL_LANES=$r
done
TREE=wide run "wide blocks 1536 no staged nodes" ETX_HIP_LIBRARY=$V/libetx_hip_dbg.so ETX_HIP_DEBUG_BLOCKS=1536 ETX_HIP_LDS_NODES=0
for b in 1024 1536; do
  v=$(ETX_HIP_LIBRARY=$V/libetx_hip_dbg.so ETX_HIP_DEBUG_BLOCKS=$b timeout 300 python bench.py --workload sssdragon_bdpt --bvh wide --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/sweep.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "sssdragon_bdpt wide blocks $b: $v" >> $O/sweep.txt
done
for l in 1 2 3 4; do
  ETX_HIP_LANES=$l timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/sweep.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full lanes $l:', d['value'], 'Msamples/s, working set', d['config']['working_set_gb'], 'GB')" >> $O/sweep.txt
done
cat $O/sweep.txt; tail -5 $O/sweep.err
