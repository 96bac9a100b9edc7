#!/bin/bash
# Round 4, call l: the round's profiles of the final build - rocprofv3 kernel statistics (default lanes) and six one-lane PMC passes per workload
# (tools/profile_round.sh), for configs[1], configs[3], configs[4]; then the bench lines.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for w in full sssdragon_bdpt cloud_bdpt; do
  bash tools/profile_round.sh r4_$w --workload $w > gpurun_out/profile_$w.log 2>&1
done
O=gpurun_out/r4l
mkdir -p $O
for w in full sssdragon_bdpt cloud_bdpt; do
  d=gpurun_out/prof_r4_$w
  cp $d/pmc_summary.json $O/round4_pmc_${w}_1lane_summary.json 2>/dev/null
  cp $d/pmc_summary.txt $O/round4_pmc_${w}_1lane_summary.txt 2>/dev/null
  cp $d/kernel_stats.csv $O/round4_bench_${w}_kernel_stats.csv 2>/dev/null
  tail -1 $d/bench_stats.json > $O/round4_bench_${w}_under_rocprof.json 2>/dev/null
done
ls -la $O; head -5 $O/round4_pmc_full_1lane_summary.txt; python3 -c "
import json
for w in ('full','sssdragon_bdpt','cloud_bdpt'):
    try:
        d=json.load(open('$O/round4_pmc_%s_1lane_summary.json'%w)); print(w, d.get('_meta'))
    except Exception as e: print(w, 'missing', e)
"
