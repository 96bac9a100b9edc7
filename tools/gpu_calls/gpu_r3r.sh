#!/bin/bash
# round 3, GPU call r: the 4096-spp cases not run since the merge histogram / shadow kernel changes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3r
mkdir -p $O
timeout 250 python -m pytest tests/test_gpu_parity_hi.py -x -q -m gpu -s -k "rough or glass or sss" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
