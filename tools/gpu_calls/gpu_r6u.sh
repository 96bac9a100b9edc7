#!/bin/bash
# round 6, GPU call u (closing, second): after the fix of k_merge_eval_generic's fold and 256 walk workgroups: (1) the driver's suite command, whole; (2) the reproducibility test that
# caught the fold, 20 fresh processes; (3) kernel statistics + one-lane counter passes + bench lines of the four workloads on the FINAL library (profiles/ refreshed on the box first).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6u
mkdir -p $O
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/tests_full.log 2>&1
echo "driver's suite rc=$? $(($(date +%s) - t0)) s: $(grep -E 'passed|failed|error' $O/tests_full.log | tail -1)" >> $O/log.txt
bad=0
for i in $(seq 1 20); do
  timeout 120 python3 -m pytest tests/test_gpu_pixel_sharding.py -x -q -m gpu -p no:cacheprovider -k "reproducible" > $O/repro_$i.log 2>&1 || bad=$((bad + 1))
done
echo "reproducibility of the rough box: 20 processes, $bad failed" >> $O/log.txt
for spec in "full:" "gems:--workload gems" "sssdragon_bdpt:--workload sssdragon_bdpt" "cloud_bdpt:--workload cloud_bdpt"; do
  w=${spec%%:*}; args=${spec#*:}
  bash tools/profile_round.sh r6u_$w $args > $O/profile_$w.log 2>&1
  d=gpurun_out/prof_r6u_$w
  cp $d/pmc_summary.json $O/round6_pmc_${w}_1lane_summary.json 2>/dev/null
  cp $d/pmc_summary.txt $O/round6_pmc_${w}_1lane_summary.txt 2>/dev/null
  cp $d/kernel_stats.csv $O/round6_bench_${w}_kernel_stats.csv 2>/dev/null
  cp $d/bench_stats.json $O/round6_bench_${w}_under_rocprof.json 2>/dev/null
  cp $d/pmc_summary.json profiles/round6_pmc_${w}_1lane_summary.json 2>/dev/null
  echo "profile $w: $(ls $d 2>/dev/null | wc -l) files" >> $O/log.txt
done
timeout 400 python3 bench.py 2>$O/bench_full.err | grep '^{' > $O/round6_bench_full_1080p.json
ETX_HIP_LANES=1 timeout 400 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $O/round6_bench_full_1lane.json
timeout 400 python3 bench.py --workload gems 2>/dev/null | grep '^{' > $O/round6_bench_gems.json
timeout 600 python3 bench.py --workload sssdragon_bdpt 2>/dev/null | grep '^{' > $O/round6_bench_sssdragon_bdpt.json
timeout 600 python3 bench.py --workload cloud_bdpt 2>/dev/null | grep '^{' > $O/round6_bench_cloud_bdpt.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_full_driver_command.json 2> $O/bench_driver_command.err
for f in $O/round6_bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], d['value'], d.get('repeats',{}).get('values'), 'stale', d.get('counters_stale'), 'roofline', d['roofline'].get('frac'), d['roofline'].get('traffic'), 'dominant', d.get('dominant_kernel',{}).get('group'))
except Exception as e:
    print('$f', 'unreadable', e)
" >> $O/log.txt; done
cat $O/log.txt
