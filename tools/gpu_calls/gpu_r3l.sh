#!/bin/bash
# round 3, GPU call l: photon grid allocated on first VCM use - more lanes for the bidirectional workloads; closing bench lines + kernel statistics
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3l
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bdpt.py tests/test_gpu_checkpoint.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
for l in 4 6; do
  for w in sssdragon_bdpt cloud_bdpt gems; do
    ETX_HIP_LANES=$l timeout 300 python bench.py --workload $w --no-cpu-baseline --no-kernel-table > $O/lanes${l}_$w.json 2> $O/lanes${l}_$w.err
    echo "lanes $l $w rc=$? $(python -c "import json,sys; print(json.load(open('$O/lanes${l}_$w.json'))['value'])" 2>/dev/null)" >> $O/log.txt
  done
done
ETX_HIP_LANES=8 timeout 300 python bench.py --workload cloud_bdpt --no-cpu-baseline --no-kernel-table > $O/lanes8_cloud_bdpt.json 2> $O/lanes8_cloud_bdpt.err
echo "lanes 8 cloud rc=$? $(python -c "import json,sys; print(json.load(open('$O/lanes8_cloud_bdpt.json'))['value'])" 2>/dev/null)" >> $O/log.txt
for w in full gems gems1m; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$? $(python -c "import json,sys; print(json.load(open('$O/bench_$w.json'))['value'])" 2>/dev/null)" >> $O/log.txt
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_full -- python $OLDPWD/bench.py --no-cpu-baseline --no-kernel-table > $OLDPWD/$O/prof_full.json 2> $OLDPWD/$O/prof_full.err )
echo "rocprof full rc=$?" >> $O/log.txt
find $O/prof_full -name "*kernel_stats.csv" -exec cp {} $O/full_kernel_stats.csv \;
find $O/prof_full -type f ! -name "*stats.csv" -delete
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
