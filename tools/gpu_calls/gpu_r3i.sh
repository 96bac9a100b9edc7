#!/bin/bash
# round 3, GPU call i: walks of a pass's tail run to their end in one launch; rocprofv3 statistics + PMC passes of configs[3] and configs[4]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3i
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest "tests/test_gpu_bdpt.py::test_bdpt_subsurface_walk_matches_reference" tests/test_gpu_sssmesh.py tests/test_gpu_parity_size.py -q -m gpu -s -k "sss or config3" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
timeout 600 python bench.py --workload sssdragon_bdpt > $O/bench_sssdragon_bdpt.json 2> $O/bench_sssdragon_bdpt.err
echo "bench sssdragon rc=$?" >> $O/log.txt
timeout 900 bash tools/profile_round.sh r3_sssdragon --workload sssdragon_bdpt > $O/profile_sssdragon.log 2>&1
echo "profile sssdragon rc=$?" >> $O/log.txt
timeout 900 bash tools/profile_round.sh r3_cloud --workload cloud_bdpt > $O/profile_cloud.log 2>&1
echo "profile cloud rc=$?" >> $O/log.txt
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
