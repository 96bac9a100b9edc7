#!/bin/bash
# round 3, GPU call n: merge histogram in the vertex store, lean shadow kernel for opaque tree scenes (A/B against the previous commit), CB box at 4096 spp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3n
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_hi.py tests/test_gpu_parity.py tests/test_gpu_sssmesh.py -x -q -m gpu -s -k "ssscb or full or gems or classic or merging or sssmesh or meshes or subsurface" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
bash tools/ab_bench.sh full 2 base head >> $O/ab.txt 2>&1
bash tools/ab_bench.sh sssdragon_bdpt 2 base noopaque opaque0 >> $O/ab.txt 2>&1
bash tools/ab_bench.sh gems 2 base noopaque opaque0 head >> $O/ab.txt 2>&1
bash tools/ab_bench.sh gems1m 1 base noopaque opaque0 >> $O/ab.txt 2>&1
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt $O/ab.txt
