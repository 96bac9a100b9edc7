#!/bin/bash
# round 3, GPU call j: medium boundaries crossed inside the flat trace kernel (VCM / BDPT) - parity + bench; PMC passes of configs[3] / configs[4]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3j
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest "tests/test_gpu_parity_hi.py::test_vcm_matches_reference_at_4096_spp" "tests/test_gpu_bdpt.py::test_bdpt_full_matches_reference_at_4096_spp" tests/test_gpu_parity_size.py "tests/test_gpu_parity.py::test_feature_scenes_match_reference" -q -m gpu -s -k "full or cloud or config1 or config4 or feature" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
for w in full cloud_bdpt sssdragon_bdpt; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$?" >> $O/log.txt
done
ETX_HIP_LANES=1 timeout 600 python bench.py --workload full --no-cpu-baseline > $O/bench_full_1lane.json 2> $O/bench_full_1lane.err
timeout 900 bash tools/profile_round.sh r3_sssdragon --workload sssdragon_bdpt > $O/profile_sssdragon.log 2>&1
echo "profile sssdragon rc=$?" >> $O/log.txt
timeout 900 bash tools/profile_round.sh r3_cloud --workload cloud_bdpt > $O/profile_cloud.log 2>&1
echo "profile cloud rc=$?" >> $O/log.txt
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
