#!/bin/bash
# round 6, GPU call s: clearance grids of subsurface materials (dev_scene.h DSssGrid): a walk event whose flight is shorter than the clearance of its position skips
# the material-filtered closest-hit query; the walk kernels handle up to E such events per trip before the one event that may traverse.
#   parity: every subsurface comparison (PT / VCM / BDPT, boxes and meshes, textured, Christensen-Burley); A/B on configs[3]: before (HEAD of the closing profiles),
#   E = 0 (plain skip), 1, 2 (product build), 3.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6s
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "sss or subsurface or sssmesh or split or walk or burley or scene_update" > $O/tests_sss.log 2>&1
echo "subsurface comparisons rc=$? $(grep -E 'passed|failed|error' $O/tests_sss.log | tail -1)" >> $O/log.txt
V=$PWD/etx-tracer_amd/variants
for r in 1 2; do
  for tag in before walk_e0 walk_e1 base walk_e3; do
    L=$V/libetx_hip_$tag.so; [ $tag = base ] && L=$PWD/etx-tracer_amd/libetx_hip.so
    x=$(ETX_HIP_LIBRARY=$L timeout 300 python3 bench.py --workload sssdragon_bdpt --steps 8 --warmup 4 --repeats 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], 'rounds', d['counters']['wavefront_rounds_per_step'])")
    echo "sssdragon_bdpt $tag run $r: $x" >> $O/ab_clearance.txt
  done
done
for tag in before base; do
  L=$V/libetx_hip_$tag.so; [ $tag = base ] && L=$PWD/etx-tracer_amd/libetx_hip.so
  x=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$L timeout 400 python3 bench.py --workload sssdragon_bdpt --steps 4 --warmup 2 --repeats 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if isinstance(v, dict)}, 'upload_s', d['config']['tree'].get('upload_s'))")
  echo "sssdragon_bdpt 1 lane $tag: $x" >> $O/ab_clearance.txt
done
cat $O/log.txt $O/ab_clearance.txt
