#!/bin/bash
# round 5, GPU call d: the whole GPU suite as the driver runs it (-x), now with the two halves of every high-sample-count comparison rendered concurrently by
# two contexts; the printed comparisons of the option-set tests; the headline bench.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5d
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
timeout 600 python -m pytest tests/test_gpu_options.py -q -m gpu -s > $O/option_tests.log 2>&1
echo "option tests rc=$?" >> $O/log.txt
grep -h "block-8 RMSE" $O/option_tests.log | sed 's/^\.*//' > $O/round5_option_tests.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_full.json
echo "full $(python -c "import json; d=json.load(open('$O/bench_full.json')); print(d['value'], d['repeats']['values'])" 2>/dev/null)" >> $O/log.txt
tail -n 25 $O/tests.log | grep -v "^$"
tail -n 3 $O/option_tests.log
cat $O/log.txt
