#!/bin/bash
# round 6, GPU call v: the driver's suite command on the final library once more (a second fresh box), then smoke() and the driver's bench command line.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6v
mkdir -p $O
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/tests_full.log 2>&1
echo "driver's suite rc=$? $(($(date +%s) - t0)) s: $(grep -E 'passed|failed|error' $O/tests_full.log | tail -1)" >> $O/log.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$? $(grep smoke: $O/smoke.log | tail -1)" >> $O/log.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['repeats']['values'], 'stale', d['counters_stale'], 'library', d['config']['library_sha16'])" >> $O/log.txt
cat $O/log.txt
