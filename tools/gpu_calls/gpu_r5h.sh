#!/bin/bash
# round 5, GPU call h (closing): configs[2] at size against the 128-spp reference film (first run: prints what the limits are set from), the whole GPU suite of
# the tree as shipped, the headline bench, and the film reduce's device time on the 2048 x 2048 bidirectional workload (four layers, 268 MB).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5h
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity_size.py -q -m gpu -s -k "config2 or config1_full_1080p_matches_reference" > $O/tests_size.log 2>&1
echo "size tests rc=$?" >> $O/log.txt
grep "rel mean" $O/tests_size.log >> $O/log.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
echo "driver command: last stdout line is JSON: $(tail -n 1 $O/bench_driver_command.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['values'], d['counters_stale'])" 2>&1 | tail -1)" >> $O/log.txt
timeout 300 python bench.py --comm-single --no-cpu-baseline --no-kernel-table > $O/bench_comm_single_stdout.txt 2>/dev/null
echo "comm-single: last stdout line is JSON: $(tail -n 1 $O/bench_comm_single_stdout.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['reduce'])" 2>&1 | tail -1)" >> $O/log.txt
timeout 400 python bench.py --workload cloud_bdpt --comm-single --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' > $O/bench_cloud_comm_single.json
echo "cloud comm-single $(python -c "import json; d=json.load(open('$O/bench_cloud_comm_single.json')); print(d['value'], d['repeats']['values'], d['reduce'])" 2>&1 | tail -1)" >> $O/log.txt
timeout 400 python bench.py --workload cloud_bdpt --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' > $O/bench_cloud_plain.json
echo "cloud plain $(python -c "import json; d=json.load(open('$O/bench_cloud_plain.json')); print(d['value'], d['repeats']['values'])" 2>&1 | tail -1)" >> $O/log.txt
tail -n 22 $O/tests.log | grep -v "^$"
cat $O/log.txt
