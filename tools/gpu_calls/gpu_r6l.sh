#!/bin/bash
# round 6, GPU call l: the bidirectional kernels of mixed scenes split their items by BSDF class (kPartSimple / kPartGeneral, kernels_bdpt.hip).
#   parity: every bidirectional comparison + the new split-vs-unsplit test; A/B on configs[3] (sssdragon_bdpt): --debug-flags 0x10000 = unsplit, interleaved;
#   per-group times of both on one lane.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6l
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_bdpt.py tests/test_gpu_sssmesh.py tests/test_gpu_pixel_sharding.py -x -q -m gpu -p no:cacheprovider > $O/tests_bdpt.log 2>&1
echo "bdpt + sssmesh + pixel sharding rc=$? $(grep -E 'passed|failed|error' $O/tests_bdpt.log | tail -1)" >> $O/log.txt
timeout 900 python3 -m pytest tests/test_gpu_parity_size.py tests/test_gpu_options.py -x -q -m gpu -p no:cacheprovider -k "bdpt or bidirectional or sssdragon or cloud" > $O/tests_size.log 2>&1
echo "size + options (bidirectional) rc=$? $(grep -E 'passed|failed|error' $O/tests_size.log | tail -1)" >> $O/log.txt
for r in 1 2; do
  for flags in 0 65536; do
    x=$(timeout 400 python3 bench.py --workload sssdragon_bdpt --steps 8 --warmup 4 --repeats 3 --debug-flags $flags --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
    echo "sssdragon_bdpt flags $flags run $r: $x" >> $O/ab_split.txt
  done
done
for flags in 0 65536; do
  x=$(ETX_HIP_LANES=1 timeout 400 python3 bench.py --workload sssdragon_bdpt --steps 4 --warmup 2 --repeats 3 --debug-flags $flags --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items() if isinstance(v, dict)})")
  echo "sssdragon_bdpt 1 lane flags $flags: $x" >> $O/ab_split.txt
done
cat $O/log.txt $O/ab_split.txt
