#!/bin/bash
# Round 4, call v: the whole GPU suite on the build with pixel sharding, medium rows, the 32-entry path table and the resident-workgroup shade grids.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4v
mkdir -p $O
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -q -m gpu --durations=25 > $O/tests.log 2>&1
echo "suite rc=$?" > $O/log.txt
grep -v "^  File\|^Extension" $O/tests.log | tail -45; cat $O/log.txt
