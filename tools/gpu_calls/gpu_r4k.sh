#!/bin/bash
# Round 4, call k: the whole GPU suite again (pair lists with holes skipped; pre-scheduled passes), then scheduled vs polled passes A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4k
mkdir -p $O
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -q -m gpu -x --durations=30 > $O/tests.log 2>&1
echo "suite rc=$?" > $O/log.txt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['counters']['wavefront_rounds_per_step'])"; }
V=etx-tracer_amd/variants/libetx_hip_dbgapi.so
for round in 1 2; do
  for s in 1 0; do
    for l in 4 1; do
      r=$(ETX_HIP_LIBRARY=$V ETX_HIP_SCHEDULED_PASSES=$s ETX_HIP_LANES=$l timeout 120 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
      echo "full scheduled=$s lanes=$l: $r" >> $O/log.txt
    done
  done
done
grep -v "^  File\|^Extension" $O/tests.log | tail -45; cat $O/log.txt
