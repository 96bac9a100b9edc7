#!/bin/bash
# round 3, GPU call m: lanes per integrator (four / six for the bidirectional one), pools of the extra lanes and the photon grid on first use
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3m
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_bdpt.py tests/test_gpu_binding.py tests/test_gpu_scene_update.py tests/test_gpu_repeated_render.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
for w in full cloud_bdpt sssdragon_bdpt; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$? $(python -c "import json,sys; d=json.load(open('$O/bench_$w.json')); print(d['value'], d['config']['working_set_gb'])" 2>/dev/null)" >> $O/log.txt
done
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
