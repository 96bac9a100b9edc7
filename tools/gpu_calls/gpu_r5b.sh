#!/bin/bash
# round 5, GPU call b: the option-set tests again (classic-box films with independent streams, heavy-tail limits), the reduce that never waits
# (bench --comm-single), the phased tree traversal kernel (ETX_HIP_BVH_VARIANT=3 in a debug build of kernels_trace.hip): hits against the brute-force
# oracle and the device-built tree, rate alone, counters alone, gems pipeline.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_options.py tests/test_gpu_binding.py::test_cpp_binding_reference_seeding_option tests/test_gpu_parity.py::test_rccl_film_reduce_cadence_single_rank \
  tests/test_gpu_parity.py::test_pool_overflow_is_reported_and_does_not_skip_the_reduce tests/test_gpu_parity.py::test_pools_grow_and_the_overflowed_iteration_is_rendered_again \
  tests/test_gpu_parity.py::test_two_shards_with_uneven_iteration_counts tests/test_gpu_pixel_sharding.py -q -m gpu -s --durations=10 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
P=$PWD/etx-tracer_amd/variants/libetx_hip_phased.so
ETX_HIP_LIBRARY=$P ETX_HIP_BVH_VARIANT=3 timeout 600 python -m pytest tests/test_gpu_parity.py -k "traversal or alpha" tests/test_gpu_scene_update.py -q -m gpu > $O/tests_phased.log 2>&1
echo "tests phased rc=$?" >> $O/log.txt
for r in 1 2; do
  for v in 0 3; do
    echo "== gems alone, variant $v, run $r" >> $O/trace_bench.txt
    ETX_HIP_LIBRARY=$P ETX_HIP_BVH_VARIANT=$v timeout 200 python tools/trace_bench.py tests/golden/cornell_gems_1080p.etxscene 2073600 20 2>/dev/null | grep rays >> $O/trace_bench.txt
  done
done
for r in 1 2; do
  for v in 0 3; do
    x=$(ETX_HIP_LIBRARY=$P ETX_HIP_BVH_VARIANT=$v timeout 300 python bench.py --workload gems --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], 'trace frac', d['roofline']['frac'])")
    echo "gems pipeline variant $v run $r: $x" >> $O/trace_bench.txt
  done
done
( export ETX_HIP_LIBRARY=$P ETX_HIP_BVH_VARIANT=3; timeout 500 tools/profile_trace_alone.sh gems4_phased $PWD/tests/golden/cornell_gems_1080p.etxscene > $O/trace_alone_gems4_phased.txt 2>&1 )
cp -r gpurun_out/trace_alone_gems4_phased $O/ 2>/dev/null
timeout 300 python bench.py --comm-single --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' > $O/bench_comm_single.json
echo "comm_single $(python -c "import json; d=json.load(open('$O/bench_comm_single.json')); print(d['value'], d['repeats']['values'], d['reduce'])" 2>/dev/null)" >> $O/log.txt
timeout 300 python bench.py --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' > $O/bench_plain.json
echo "plain $(python -c "import json; d=json.load(open('$O/bench_plain.json')); print(d['value'], d['repeats']['values'])" 2>/dev/null)" >> $O/log.txt
timeout 300 python bench.py --comm-single --reduce-every 4 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' > $O/bench_comm_single_every4.json
echo "comm_single every 4 $(python -c "import json; d=json.load(open('$O/bench_comm_single_every4.json')); print(d['value'], d['repeats']['values'], d['reduce'])" 2>/dev/null)" >> $O/log.txt
grep -n "passed\|failed" $O/tests.log $O/tests_phased.log | tail -n 6
cat $O/log.txt $O/trace_bench.txt
