#!/bin/bash
# Round 4, call b: the new parity cases (defaults with blue noise at 4096 spp; shared streams against the pinned reference) and a sweep of the
# tree traversal kernel's launch shape (debug variants of kernels_trace.hip read the ETX_HIP_* knobs) on both trees.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity_hi.py -q -m gpu -s -k "blue_noise_at_4096 or shared_streams" > $O/hi_tests.log 2>&1
echo "hi tests rc=$?" >> $O/log.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "pool" > $O/pool_tests.log 2>&1
echo "pool tests rc=$?" >> $O/log.txt
for w in full sssdragon_bdpt cloud_bdpt; do
  timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-table 2>$O/bench_$w.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], 'Msamples/s, working set', d['config']['working_set_gb'], 'GB')" >> $O/log.txt
done
S=tests/golden/cornell_gems_1080p.etxscene
run() { # label env... -- tree
  label=$1; shift
  echo "== $label" >> $O/sweep.txt
  env "$@" timeout 120 python tools/trace_bench.py $S 2073600 20 $TREE 2>/dev/null | grep -v amdgpu.ids >> $O/sweep.txt
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
  v=$(ETX_HIP_LIBRARY=$V/libetx_hip_dbg.so ETX_HIP_DEBUG_BLOCKS=$b timeout 300 python bench.py --workload sssdragon_bdpt --bvh wide --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "sssdragon_bdpt wide blocks $b: $v" >> $O/sweep.txt
done
tail -n 25 $O/hi_tests.log; tail -n 5 $O/pool_tests.log
cat $O/log.txt $O/sweep.txt
