#!/bin/bash
# round 5, GPU call e (closing): the round's profiles of the FINAL library - rocprofv3 kernel statistics (default lanes) and six one-lane PMC passes per
# workload (tools/profile_round.sh) for configs[1], configs[3], configs[4] - then the bench lines of every workload, which now find counters of their own
# library (counters_stale false), and the reduce's cost on a one-rank communicator.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
for w in full sssdragon_bdpt cloud_bdpt; do
  bash tools/profile_round.sh r5_$w --workload $w > gpurun_out/profile_r5_$w.log 2>&1
  d=gpurun_out/prof_r5_$w
  cp $d/pmc_summary.json $O/round5_pmc_${w}_1lane_summary.json 2>/dev/null
  cp $d/pmc_summary.txt $O/round5_pmc_${w}_1lane_summary.txt 2>/dev/null
  cp $d/kernel_stats.csv $O/round5_bench_${w}_kernel_stats.csv 2>/dev/null
  tail -1 $d/bench_stats.json > $O/round5_bench_${w}_under_rocprof.json 2>/dev/null
  # the bench line must find the summary of ITS library: copy it where tools/profile_lookup.py looks
  cp $d/pmc_summary.json profiles/round5_pmc_${w}_1lane_summary.json 2>/dev/null
done
timeout 400 python bench.py 2>$O/full.err | grep '^{' > $O/round5_bench_full_1080p.json
for w in classic gems sssdragon_bdpt cloud_bdpt; do
  timeout 500 python bench.py --workload $w 2>$O/$w.err | grep '^{' > $O/round5_bench_$w.json
done
timeout 300 python bench.py --comm-single --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' > $O/round5_bench_full_comm_single.json
ETX_HIP_LANES=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $O/round5_bench_full_1lane.json
for f in $O/round5_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], d['value'], d['repeats']['values'], 'ms', d['ms_per_step'], 'ws', d['config']['working_set_gb'], 'grows', d['config']['pool_grows'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), 'stale', d.get('counters_stale'), 'dominant', d['dominant_kernel']['group'] if d.get('dominant_kernel') else None, d['dominant_kernel']['frac'] if d.get('dominant_kernel') else None, 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'reduce', d.get('reduce'))
except Exception as e: print('$f', 'unreadable', e)
"; done
ls $O
