#!/bin/bash
# round 3, GPU call d: simple-BSDF instantiations of the bidirectional kernels + pair classes, checkpoint / adaptive / comm fixes, spin back-off
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3d
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bdpt.py tests/test_gpu_sssmesh.py tests/test_gpu_checkpoint.py "tests/test_gpu_parity.py::test_pt_adaptive_sampling" tests/test_gpu_parity_size.py -x -q -m gpu -s -k "not config2 and not config1_full_1080p_matches_reference" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
for w in full cloud_bdpt sssdragon_bdpt; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$?" >> $O/log.txt
done
for w in sssdragon_bdpt cloud_bdpt; do
  ( cd /tmp && ETX_HIP_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$w.json 2> $GRAFT_REPO_ROOT/$O/prof_$w.err )
  echo "prof $w rc=$?" >> $O/log.txt
  find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_1lane_kernel_stats.csv \;
done
tail -n 6 $O/tests.log
cat $O/log.txt
