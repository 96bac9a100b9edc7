#!/bin/bash
# round 3, second GPU call: the walk queue (k_bdpt_walk_*) - parity of the subsurface BDPT tests, configs[3]/[4] at size, bench + kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3c
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bdpt.py tests/test_gpu_sssmesh.py "tests/test_gpu_parity.py::test_pt_adaptive_sampling" -x -q -m gpu -s > $O/test_bdpt.log 2>&1
echo "tests bdpt rc=$?" >> $O/log.txt
timeout 600 python bench.py --workload sssdragon_bdpt --steps 8 --warmup 2 > $O/bench_sssdragon_bdpt.json 2> $O/bench_sssdragon_bdpt.err
echo "bench sssdragon rc=$?" >> $O/log.txt
ETX_HIP_LANES=1 timeout 600 python bench.py --workload sssdragon_bdpt --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_sssdragon_bdpt_1lane.json 2> $O/bench_sssdragon_bdpt_1lane.err
timeout 900 python -m pytest tests/test_gpu_parity_size.py -x -q -m gpu -s -k "config3 or config4 or unmodified" > $O/test_size.log 2>&1
echo "tests size rc=$?" >> $O/log.txt
for w in sssdragon_bdpt cloud_bdpt; do
  ( cd /tmp && ETX_HIP_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$w.json 2> $GRAFT_REPO_ROOT/$O/prof_$w.err )
  echo "prof $w rc=$?" >> $O/log.txt
  find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_1lane_kernel_stats.csv \;
done
tail -n 5 $O/test_bdpt.log; tail -n 8 $O/test_size.log
cat $O/log.txt
