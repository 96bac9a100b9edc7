#!/bin/bash
# round 5, GPU call i: counters of the gems workload on the final library (so that `bench.py --workload gems` quotes its own), and a longer headline run
# (five regions of 64 steps) for the spread of one box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
bash tools/profile_round.sh r5_gems --workload gems > gpurun_out/profile_r5_gems.log 2>&1
d=gpurun_out/prof_r5_gems
cp $d/pmc_summary.json $O/round5_pmc_gems_1lane_summary.json 2>/dev/null
cp $d/pmc_summary.txt $O/round5_pmc_gems_1lane_summary.txt 2>/dev/null
cp $d/kernel_stats.csv $O/round5_bench_gems_kernel_stats.csv 2>/dev/null
cp $d/pmc_summary.json profiles/round5_pmc_gems_1lane_summary.json 2>/dev/null
timeout 400 python bench.py --workload gems 2>/dev/null | grep '^{' > $O/round5_bench_gems.json
timeout 400 python bench.py --steps 64 --warmup 8 --repeats 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/round5_bench_full_5x64.json
for f in $O/round5_bench_gems.json $O/round5_bench_full_5x64.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], d['value'], d['repeats'], 'stale', d['counters_stale'], 'dominant', d['dominant_kernel']['group'], d['dominant_kernel'].get('counters_1lane'))
"; done
ls $O
