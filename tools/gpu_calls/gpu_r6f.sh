#!/bin/bash
# round 6, GPU call f: first run of this round's kernel work.
#   1. matrix-core flat sweep (debug flag 128) against the VALU sweep: hits compared, kernel alone timed (tools/trace_bench.py), pipeline A/B (bench.py --debug-flags)
#   2. the split merge of generic materials (k_merge_filter_generic + k_merge_eval_generic): gems / rough / glass parity tests, gems bench
#   3. new tests of this round (runtime, contexts)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6f
mkdir -p $O
export TMPDIR=/tmp
for scene in full classic; do
  timeout 300 python3 tools/trace_bench.py tests/golden/cornell_${scene}_1080p.etxscene 2073600 20 --flags 0,64,128 --check > $O/trace_bench_$scene.txt 2>&1
  echo "trace_bench $scene rc=$?" >> $O/log.txt
done
timeout 900 python3 -m pytest tests/test_gpu_runtime.py tests/test_gpu_contexts.py -x -q -m gpu -p no:cacheprovider > $O/tests_new.log 2>&1
echo "new tests rc=$? $(tail -1 $O/tests_new.log)" >> $O/log.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_repeated_render.py -x -q -m gpu -p no:cacheprovider > $O/tests_parity.log 2>&1
echo "parity + repeated render rc=$? $(tail -1 $O/tests_parity.log)" >> $O/log.txt
timeout 900 python3 -m pytest tests/test_gpu_parity_hi.py -x -q -m gpu -p no:cacheprovider -k "gems or rough or glass" > $O/tests_hi_generic.log 2>&1
echo "4096-spp generic-material films rc=$? $(tail -1 $O/tests_hi_generic.log)" >> $O/log.txt
for r in 1 2; do
  for flags in 0 128; do
    for w in full classic; do
      x=$(timeout 300 python3 bench.py --workload $w --steps 24 --warmup 6 --repeats 3 --no-cpu-baseline --no-kernel-table --debug-flags $flags 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], 'trace avg launch ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])")
      echo "$w flags $flags run $r: $x" >> $O/ab_mfma_pipeline.txt
    done
  done
done
timeout 600 python3 bench.py --workload gems --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_gems.json 2> $O/bench_gems.err
echo "bench gems rc=$? $(python3 -c "import json; d=json.loads(open('$O/bench_gems.json').read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], {k: (v['ms_per_step'], v['share']) for k, v in d['kernels'].items() if isinstance(v, dict)})")" >> $O/log.txt
ETX_HIP_LANES=1 timeout 600 python3 bench.py --workload gems --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_gems_1lane.json 2>> $O/bench_gems.err
echo "bench gems 1 lane rc=$? $(python3 -c "import json; d=json.loads(open('$O/bench_gems_1lane.json').read().strip().splitlines()[-1]); print(d['value'], {k: (v['ms_per_step'], v['share']) for k, v in d['kernels'].items() if isinstance(v, dict)})")" >> $O/log.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
echo "bench full rc=$? $(python3 -c "import json; d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]); print(d['value'], d['repeats'], d['config']['runtime'])")" >> $O/log.txt
cat $O/log.txt $O/ab_mfma_pipeline.txt; cat $O/trace_bench_full.txt $O/trace_bench_classic.txt
