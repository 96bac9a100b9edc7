#!/bin/bash
# round 6, GPU call m: sweep of the walk kernels' two constants on configs[3] (events a walk gets per round: 16 / 32 (product) / 64 / 128; idle lanes that trigger a
# refill: 8 / 16 (product) / 32), experiment builds of kernels_bdpt.hip (tools/build_variant.sh), interleaved with the product library, two rounds.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6m
mkdir -p $O
export TMPDIR=/tmp
for r in 1 2; do
  for tag in base walk_b16 walk_b64 walk_b128 walk_r8 walk_r32; do
    L=$PWD/etx-tracer_amd/variants/libetx_hip_$tag.so; [ $tag = base ] && L=$PWD/etx-tracer_amd/libetx_hip.so
    x=$(ETX_HIP_LIBRARY=$L timeout 300 python3 bench.py --workload sssdragon_bdpt --steps 8 --warmup 4 --repeats 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], 'rounds', d['counters']['wavefront_rounds_per_step'])")
    echo "sssdragon_bdpt $tag run $r: $x" >> $O/walk_sweep.txt
  done
done
cat $O/walk_sweep.txt
# kernel statistics + the one-lane counter passes of configs[3] on the product library (the split kernels): gpurun_out/prof_r6_sssdragon
bash tools/profile_round.sh r6_sssdragon --workload sssdragon_bdpt > $O/profile_round.log 2>&1
tail -3 $O/profile_round.log
