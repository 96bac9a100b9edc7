#!/bin/bash
# round 5, GPU call f: (1) the subsurface-box comparisons at 1024 spp (tests changed after call d); (2) same-box A/B of the tree workloads: the library of
# the round's first commit (interleaved tree kernel, eight-wide format still in, built from `git archive 967213d`) against the final one, interleaved.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5f
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest "tests/test_gpu_parity_hi.py::test_vcm_matches_reference_at_4096_spp[sss]" "tests/test_gpu_parity_hi.py::test_pt_matches_reference_at_4096_spp[sss]" \
  "tests/test_gpu_parity_hi.py::test_vcm_matches_reference_at_4096_spp[classic]" -q -m gpu -s --durations=5 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
OLD=$PWD/etx-tracer_amd/variants/libetx_hip_r5first.so
for r in 1 2 3; do
  for w in sssdragon_bdpt gems cloud_bdpt; do
    for lib in old new; do
      L=$PWD/etx-tracer_amd/libetx_hip.so; [ $lib = old ] && L=$OLD
      x=$(ETX_HIP_LIBRARY=$L timeout 300 python bench.py --workload $w --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
      echo "$w $lib run $r: $x" >> $O/ab_tree_workloads.txt
    done
  done
done
grep -n "passed\|failed" $O/tests.log | tail -3
grep "block-8" $O/tests.log | cut -c1-220
cat $O/log.txt $O/ab_tree_workloads.txt
