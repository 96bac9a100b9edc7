#!/bin/bash
# round 6, GPU call t: (1) workgroups of the persistent walk kernels (each holds 40 KB of LDS: 1024 of them = four per CU fill every CU's LDS for the walk kernel's lifetime and
# keep the other lanes' LDS-using kernels out): 256 / 512 / 1024 (product) / 2048 on configs[3], interleaved; (2) the driver's suite command on HEAD, whole (third run of the round).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6t
mkdir -p $O
export TMPDIR=/tmp
V=$PWD/etx-tracer_amd/variants
for r in 1 2; do
  for tag in base wb256 wb512 wb2048; do
    L=$V/libetx_hip_$tag.so; [ $tag = base ] && L=$PWD/etx-tracer_amd/libetx_hip.so
    x=$(ETX_HIP_LIBRARY=$L timeout 300 python3 bench.py --workload sssdragon_bdpt --steps 8 --warmup 4 --repeats 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
    echo "sssdragon_bdpt $tag run $r: $x" >> $O/ab_walk_blocks.txt
  done
done
cat $O/ab_walk_blocks.txt
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/tests_full.log 2>&1
echo "driver's suite rc=$? $(($(date +%s) - t0)) s: $(grep -E 'passed|failed|error' $O/tests_full.log | tail -1)" >> $O/log.txt
timeout 300 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1
echo "smoke rc=$? $(grep smoke: $O/smoke.log | tail -1)" >> $O/log.txt
cat $O/log.txt
