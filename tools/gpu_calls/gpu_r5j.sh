#!/bin/bash
# round 5, GPU call j: software prefetch of the next item's hit and path state in the simple group's shade kernels (-DETX_SHADE_PREFETCH,
# tools/experiments/round5_shade_prefetch.patch; 188 -> 224 / 213 -> 241 VGPRs, no spills) against the product library, interleaved.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5j
mkdir -p $O
export TMPDIR=/tmp
P=$PWD/etx-tracer_amd/variants/libetx_hip_prefetch.so
ETX_HIP_LIBRARY=$P timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vcm_classic_cornell or vcm_full_cornell or vcm_default_options" > $O/tests_prefetch.log 2>&1
echo "tests with the prefetch library rc=$? $(tail -1 $O/tests_prefetch.log)" >> $O/log.txt
for r in 1 2 3; do
  for w in full classic; do
    for lib in base prefetch; do
      L=$PWD/etx-tracer_amd/libetx_hip.so; [ $lib = prefetch ] && L=$P
      x=$(ETX_HIP_LIBRARY=$L timeout 200 python bench.py --workload $w --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
      echo "$w $lib run $r: $x" >> $O/ab_prefetch.txt
    done
  done
done
for lib in base prefetch; do
  L=$PWD/etx-tracer_amd/libetx_hip.so; [ $lib = prefetch ] && L=$P
  x=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$L timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if isinstance(v, dict)})")
  echo "full 1 lane $lib: $x" >> $O/ab_prefetch.txt
done
cat $O/log.txt $O/ab_prefetch.txt
