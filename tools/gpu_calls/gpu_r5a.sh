#!/bin/bash
# round 5, GPU call a: the new parity tests (every non-default option set against the reference, reference seeding through both bindings, the
# film reduce cadence on a one-rank communicator), fused rounds (check + A/B), hipGraph replay micro-benchmark, the tree traversal kernel alone
# under counters (four- and eight-wide), the reduce's device time at 1080p.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_options.py tests/test_gpu_binding.py::test_cpp_binding_reference_seeding_option tests/test_gpu_parity.py::test_rccl_film_reduce_cadence_single_rank \
  "tests/test_gpu_parity_hi.py::test_vcm_shared_streams_match_the_pinned_reference" "tests/test_gpu_bdpt.py::test_bdpt_shared_streams_match_the_pinned_reference" \
  tests/test_gpu_checkpoint.py -q -m gpu -s --durations=15 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_dbg.so timeout 300 python tools/fuse_check.py > $O/fuse_check.txt 2>&1
echo "fuse_check rc=$?" >> $O/log.txt
# interleaved A/B, one box: fused rounds (ETX_HIP_FUSE_TRACE) x pair order (debug flag 2048 = 0x800 = the vertex-major order of rounds 1-4)
for r in 1 2 3; do
  for w in full classic; do
    for combo in "0 2048" "0 0" "1 2048" "1 0"; do
      set -- $combo
      v=$(ETX_HIP_FUSE_TRACE=$1 ETX_HIP_DEBUG_FLAGS=$2 ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_dbg.so timeout 200 python bench.py --workload $w --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], 'trace frac', d['roofline']['frac'])")
      echo "$w fuse=$1 flags=$2 run $r: $v" >> $O/ab_fuse.txt
    done
  done
done
for combo in "0 2048" "0 0" "1 2048" "1 0"; do
  set -- $combo
  v=$(ETX_HIP_LANES=1 ETX_HIP_FUSE_TRACE=$1 ETX_HIP_DEBUG_FLAGS=$2 ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_dbg.so timeout 200 python bench.py --workload full --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], {k: (v['ms_per_step'], v['share']) for k, v in d['kernels'].items() if isinstance(v, dict)})")
  echo "full 1 lane fuse=$1 flags=$2: $v" >> $O/ab_fuse.txt
done
timeout 200 tools/micro/bin/graph_bench > $O/graph_bench.txt 2>&1
echo "graph_bench rc=$?" >> $O/log.txt
timeout 300 python bench.py --comm-single --no-cpu-baseline --no-kernel-table > $O/bench_comm_single.json 2> $O/bench_comm_single.err
echo "comm_single rc=$? $(python -c "import json; d=json.load(open('$O/bench_comm_single.json')); print(d['value'], d['reduce'])" 2>/dev/null)" >> $O/log.txt
timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > $O/bench_plain.json 2> $O/bench_plain.err
echo "plain rc=$? $(python -c "import json; d=json.load(open('$O/bench_plain.json')); print(d['value'], d['repeats'], d['reduce'])" 2>/dev/null)" >> $O/log.txt
timeout 500 tools/profile_trace_alone.sh gems4 $PWD/tests/golden/cornell_gems_1080p.etxscene > $O/trace_alone_gems4.txt 2>&1
timeout 500 tools/profile_trace_alone.sh gems8 $PWD/tests/golden/cornell_gems_1080p.etxscene wide > $O/trace_alone_gems8.txt 2>&1
cp -r gpurun_out/trace_alone_gems4 gpurun_out/trace_alone_gems8 $O/ 2>/dev/null
grep -n "passed\|failed\|error" $O/tests.log | tail -n 5
cat $O/log.txt $O/ab_fuse.txt
tail -3 $O/fuse_check.txt
