#!/bin/bash
# round 6, GPU call n: the bidirectional pair expansion writes two lists (simple / general pairs) from the paths' rows and chunks (no list walking, no per-pair classification).
#   parity: every bidirectional comparison; A/B on configs[3] and configs[4] against the library of call m (variants/libetx_hip_walk_r8.so: the code before this change, equal to the product of then within noise)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6n
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_bdpt.py tests/test_gpu_sssmesh.py tests/test_gpu_pixel_sharding.py tests/test_gpu_contexts.py -x -q -m gpu -p no:cacheprovider > $O/tests_bdpt.log 2>&1
echo "bdpt + sssmesh + pixel sharding + contexts rc=$? $(grep -E 'passed|failed|error' $O/tests_bdpt.log | tail -1)" >> $O/log.txt
timeout 900 python3 -m pytest tests/test_gpu_parity_size.py tests/test_gpu_options.py tests/test_gpu_checkpoint.py tests/test_gpu_binding.py -x -q -m gpu -p no:cacheprovider -k "bdpt or bidirectional or sssdragon or cloud or binding or checkpoint" > $O/tests_size.log 2>&1
echo "size + options + checkpoint + binding (bidirectional) rc=$? $(grep -E 'passed|failed|error' $O/tests_size.log | tail -1)" >> $O/log.txt
BEFORE=$PWD/etx-tracer_amd/variants/libetx_hip_walk_r8.so
AFTER=$PWD/etx-tracer_amd/libetx_hip.so
for r in 1 2; do
  for w in sssdragon_bdpt cloud_bdpt; do
    for which in before after; do
      L=$AFTER; [ $which = before ] && L=$BEFORE
      x=$(ETX_HIP_LIBRARY=$L timeout 400 python3 bench.py --workload $w --steps 8 --warmup 4 --repeats 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], 'pairs', d['counters']['units_per_step']['pairs'])")
      echo "$w $which run $r: $x" >> $O/ab_pair_lists.txt
    done
  done
done
for which in before after; do
  L=$AFTER; [ $which = before ] && L=$BEFORE
  x=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$L timeout 400 python3 bench.py --workload sssdragon_bdpt --steps 4 --warmup 2 --repeats 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if isinstance(v, dict)})")
  echo "sssdragon_bdpt 1 lane $which: $x" >> $O/ab_pair_lists.txt
done
cat $O/log.txt $O/ab_pair_lists.txt
