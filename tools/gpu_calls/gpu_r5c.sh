#!/bin/bash
# round 5, GPU call c: thresholds of the phased tree kernel (node-phase exit, refill) on the kernel alone, then the whole GPU suite on the tree without
# the eight-wide format, then the headline bench.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c
mkdir -p $O
export TMPDIR=/tmp
P=$PWD/etx-tracer_amd/variants/libetx_hip_phased.so
for refill in 8 16 32; do
  for phase in 8 16 24 32 48; do
    x=$(ETX_HIP_LIBRARY=$P ETX_HIP_REFILL_LANES=$refill ETX_HIP_NODE_PHASE_LANES=$phase timeout 120 python tools/trace_bench.py tests/golden/cornell_gems_1080p.etxscene 2073600 20 2>/dev/null | grep rays | awk '{print $1, $6, $7}' | tr '\n' ' ')
    echo "refill $refill node_phase $phase: $x" >> $O/phase_sweep.txt
  done
done
timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_full.json
echo "full $(python -c "import json; d=json.load(open('$O/bench_full.json')); print(d['value'], d['repeats']['values'], d['counters_stale'])" 2>/dev/null)" >> $O/log.txt
tail -n 40 $O/tests.log | grep -v "^$" | tail -n 32
cat $O/log.txt $O/phase_sweep.txt
