#!/bin/bash
# round 3, GPU call q: closing bench lines of all workloads, rocprofv3 kernel statistics of configs[1] / [3] / [4], then the GPU suite without the 4096-spp file
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3q
mkdir -p $O
export TMPDIR=/tmp
for w in full sssdragon_bdpt cloud_bdpt gems gems1m classic; do
  timeout 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$? $(python -c "import json,sys; d=json.load(open('$O/bench_$w.json')); print(d['value'], d['config'].get('working_set_gb'), d['cpu_baseline']['value'])" 2>/dev/null)" >> $O/log.txt
done
for w in full sssdragon_bdpt cloud_bdpt; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_$w -- python $OLDPWD/bench.py --workload $w --no-cpu-baseline --no-kernel-table > $OLDPWD/$O/prof_$w.json 2> $OLDPWD/$O/prof_$w.err )
  echo "rocprof $w rc=$?" >> $O/log.txt
  find $O/prof_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_kernel_stats.csv \;
  rm -rf $O/prof_$w
done
timeout 330 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bdpt.py tests/test_gpu_checkpoint.py tests/test_gpu_binding.py tests/test_gpu_sssmesh.py tests/test_gpu_parity_size.py tests/test_gpu_scene_update.py tests/test_gpu_repeated_render.py -x -q -m gpu --durations=12 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
