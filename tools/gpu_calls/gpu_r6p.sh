#!/bin/bash
# round 6, GPU call p: (1) the chunked index lists of the bidirectional light paths under stress: a knob build (tools/build_variant.sh knobs "" host_api.cpp host_scene.cpp) with
# ETX_HIP_PATH_TABLE=8 (a row holds five entries: most vertices of most paths come from chunks) and =12 runs the bidirectional comparisons with the reference;
# (2) the driver's suite command, whole, second run of the round on the product library.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6p
mkdir -p $O
export TMPDIR=/tmp
K=$PWD/etx-tracer_amd/variants/libetx_hip_knobs.so
for t in 8 12; do
  ETX_HIP_LIBRARY=$K ETX_HIP_PATH_TABLE=$t timeout 900 python3 -m pytest tests/test_gpu_bdpt.py -x -q -m gpu -p no:cacheprovider -k "classic or cloud or subsurface or split or fast" > $O/tests_table_$t.log 2>&1
  echo "path table of $t words: bidirectional comparisons rc=$? $(grep -E 'passed|failed|error' $O/tests_table_$t.log | tail -1)" >> $O/log.txt
done
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/tests_full.log 2>&1
echo "driver's suite rc=$? $(($(date +%s) - t0)) s: $(grep -E 'passed|failed|error' $O/tests_full.log | tail -1)" >> $O/log.txt
cat $O/log.txt
