#!/bin/bash
# round 3, GPU call g: after the host-side min(uint32, uint32) fix - the packed-sweep test, spectral BDPT tests, preview publishing, benches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3g
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest "tests/test_gpu_parity.py::test_two_ray_packed_sweep_matches_one_ray_sweep" "tests/test_gpu_bdpt.py::test_bdpt_spectral_scene_matches_reference" "tests/test_gpu_bdpt.py::test_bdpt_spectral_subsurface_walk_matches_reference" "tests/test_gpu_bdpt.py::test_bdpt_subsurface_walk_matches_reference" tests/test_gpu_binding.py tests/test_gpu_parity_size.py -q -m gpu -s -k "not config2 and not config1_full_1080p_matches_reference" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
for w in full cloud_bdpt sssdragon_bdpt; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$?" >> $O/log.txt
done
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
