#!/bin/bash
# round 6, GPU call i: light path head / length folded into the path's table row (one cache line per stored vertex instead of three).
#   parity subset on the new layout, then A/B against the library of call g (etx-tracer_amd/variants/libetx_hip_lazy.so: the same objects before this change)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6i
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bdpt.py tests/test_gpu_sssmesh.py -x -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1
echo "parity + bdpt + sssmesh rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)" >> $O/log.txt
BEFORE=$PWD/etx-tracer_amd/variants/libetx_hip_lazy.so
AFTER=$PWD/etx-tracer_amd/libetx_hip.so
for r in 1 2; do
  for w in full gems sssdragon_bdpt classic; do
    for which in before after; do
      L=$AFTER; [ $which = before ] && L=$BEFORE
      steps=24; [ $w = sssdragon_bdpt ] && steps=8; [ $w = gems ] && steps=12
      x=$(ETX_HIP_LIBRARY=$L timeout 400 python3 bench.py --workload $w --steps $steps --warmup 4 --repeats 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
      echo "$w $which run $r: $x" >> $O/ab_path_rows.txt
    done
  done
done
for which in before after; do
  L=$AFTER; [ $which = before ] && L=$BEFORE
  x=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$L timeout 400 python3 bench.py --workload full --steps 12 --warmup 4 --repeats 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if isinstance(v, dict)})")
  echo "full 1 lane $which: $x" >> $O/ab_path_rows.txt
done
cat $O/log.txt $O/ab_path_rows.txt
