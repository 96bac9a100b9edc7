#!/bin/bash
# round 3, GPU call k: the whole GPU suite as the driver runs it, then the bench lines of all workloads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gputest.log 2>&1
echo "gpu suite rc=$?" >> $O/log.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/log.txt
tail -n 6 $O/gputest.log
cat $O/log.txt
