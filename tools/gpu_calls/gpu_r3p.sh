#!/bin/bash
# round 3, GPU call p: event sort of the simple shade kernels (medium / surface / miss lanes of a workgroup permuted through LDS) - parity + A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3p
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_hi.py tests/test_gpu_parity.py tests/test_gpu_parity_size.py -x -q -m gpu -s -k "(vcm and (full or cloud or classic)) or heterogeneous or config1 or blue_noise or merging" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
bash tools/ab_bench.sh full 3 base nosort >> $O/ab.txt 2>&1
bash tools/ab_bench.sh classic 2 base nosort >> $O/ab.txt 2>&1
ETX_HIP_LANES=1 bash tools/ab_bench.sh full 1 base nosort >> $O/ab.txt 2>&1
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt $O/ab.txt
