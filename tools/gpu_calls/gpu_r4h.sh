#!/bin/bash
# Round 4, call h: the whole GPU suite on the build with growable pools, triangle rows and the gather-free flat transmittance walk; the default bench line.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4h
mkdir -p $O
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -q -m gpu -x --durations=25 > $O/tests.log 2>&1
echo "suite rc=$?" > $O/log.txt
timeout 200 python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
python -c "import json; d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]); print('full', d['value'], 'Msamples/s', d['config']['working_set_gb'], 'GB', d['roofline']['frac'])" >> $O/log.txt
tail -40 $O/tests.log; cat $O/log.txt
