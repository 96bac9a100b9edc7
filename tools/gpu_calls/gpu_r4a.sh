#!/bin/bash
# First GPU call of round 4: the code that has not executed on a device yet.
#   1. the eight-wide tree (csrc/dev_bvh8.h): its tests under their own timeout, so that a fault in a never-run kernel costs this step only
#   2. the 4096 / 1024-spp subsurface mesh cases (default tree) that were added after the last GPU minute
#   3. A/B of the two trees on the tree-bound workloads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
ETX_TEST_WIDE_BVH=1 timeout 300 python -m pytest tests/test_gpu_wide_bvh.py -q -m gpu -s > $O/wide_tests.log 2>&1
echo "wide tests rc=$?" >> $O/log.txt
timeout 400 python -m pytest tests/test_gpu_sssmesh.py -q -m gpu -s -k "4096 or 1024_spp" -rxX > $O/mesh_4096.log 2>&1
echo "4096-spp mesh tests rc=$?" >> $O/log.txt
for w in sssdragon_bdpt gems gems1m; do
  for b in host wide host wide; do
    v=$(timeout 300 python bench.py --workload $w --bvh $b --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-table 2>$O/bench_${w}_$b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
    echo "$w --bvh $b: $v" >> $O/ab.txt
  done
done
for t in "" wide; do
  timeout 120 python tools/trace_bench.py tests/golden/cornell_gems_1080p.etxscene 2073600 20 $t >> $O/trace_bench.txt 2>&1
done
tail -n 3 $O/wide_tests.log $O/mesh_4096.log
cat $O/log.txt $O/ab.txt $O/trace_bench.txt
