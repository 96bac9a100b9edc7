#!/bin/bash
# round 3, GPU call h: occlusion-only shadow traversal for scenes without Boundary materials, 40 KB shadow / walk kernels (3-4 workgroups per CU)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3h
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sssmesh.py "tests/test_gpu_bdpt.py::test_bdpt_spectral_scene_matches_reference" "tests/test_gpu_bdpt.py::test_bdpt_subsurface_walk_matches_reference" "tests/test_gpu_parity_hi.py::test_vcm_matches_reference_at_4096_spp" tests/test_gpu_parity_size.py -q -m gpu -s -k "(not config1_full_1080p_matches_reference) and (gems or sss or config or full or classic or cloud)" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
for w in sssdragon_bdpt gems; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$?" >> $O/log.txt
done
for w in sssdragon_bdpt; do
  ( cd /tmp && ETX_HIP_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$w.json 2> $GRAFT_REPO_ROOT/$O/prof_$w.err )
  echo "prof $w rc=$?" >> $O/log.txt
  find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_1lane_kernel_stats.csv \;
done
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt
