#!/bin/bash
# round 6, GPU call q (closing): (1) the chunked path index lists under stress (knob build, ETX_HIP_PATH_TABLE=8 / 12: rows of five / nine entries) on the bidirectional comparisons;
# (2) kernel statistics + one-lane counter passes of the four bench workloads on the FINAL library, summaries copied to profiles/ so that (3) the bench lines of the same call quote
# counters of the library they ran on; the driver's own command line last.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6q
mkdir -p $O
K=$PWD/etx-tracer_amd/variants/libetx_hip_knobs.so
for t in 8 12; do
  ETX_HIP_LIBRARY=$K ETX_HIP_PATH_TABLE=$t timeout 600 python3 -m pytest tests/test_gpu_bdpt.py -x -q -m gpu -p no:cacheprovider -k "subsurface or split or classic" > $O/tests_table_$t.log 2>&1
  echo "path table of $t words: bidirectional comparisons rc=$? $(grep -E 'passed|failed|error' $O/tests_table_$t.log | tail -1)" >> $O/log.txt
done
timeout 900 python3 -m pytest tests/test_gpu_bdpt.py tests/test_gpu_sssmesh.py tests/test_gpu_pixel_sharding.py -x -q -m gpu -p no:cacheprovider > $O/tests_bdpt.log 2>&1
echo "product library: bdpt + sssmesh + pixel sharding rc=$? $(grep -E 'passed|failed|error' $O/tests_bdpt.log | tail -1)" >> $O/log.txt
for spec in "full:" "gems:--workload gems" "sssdragon_bdpt:--workload sssdragon_bdpt" "cloud_bdpt:--workload cloud_bdpt"; do
  w=${spec%%:*}; args=${spec#*:}
  bash tools/profile_round.sh r6q_$w $args > $O/profile_$w.log 2>&1
  d=gpurun_out/prof_r6q_$w
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
timeout 300 python3 bench.py --workload classic --no-cpu-baseline 2>/dev/null | grep '^{' > $O/round6_bench_classic.json
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
