#!/bin/bash
# Round 4, call i: the pool-growth test aborted inside the suite (call h): its output uncaptured, then the suite from there on.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4i
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "pools_grow" > $O/pool.log 2>&1
echo "pool test rc=$?" > $O/log.txt
tail -30 $O/pool.log
