#!/bin/bash
# Round 4, call m: the bench lines of the round's final build (all workloads), lanes beyond four on configs[1].
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4m
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py > $O/round4_bench_full_1080p.json 2> $O/full.err
for w in classic gems gems1m sssdragon_bdpt cloud_bdpt; do
  timeout 400 python bench.py --workload $w > $O/round4_bench_$w.json 2> $O/$w.err
done
for l in 6 8; do
  ETX_HIP_LANES=$l timeout 120 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full lanes $l:', d['value'], d['config']['working_set_gb'])" >> $O/lanes.txt
done
for f in $O/round4_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], d['value'], d['unit'], 'ms', d['ms_per_step'], 'ws', d['config']['working_set_gb'], 'grows', d['config']['pool_grows'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), 'dominant', d['dominant_kernel']['group'] if d.get('dominant_kernel') else None, d['dominant_kernel']['frac'] if d.get('dominant_kernel') else None, 'cpu', (d.get('cpu_baseline') or {}).get('value'))
"; done; cat $O/lanes.txt
