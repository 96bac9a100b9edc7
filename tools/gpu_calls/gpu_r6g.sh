#!/bin/bash
# round 6, GPU call g: call f's measurements (matrix-core sweep, split merge, new tests), then the attribution matrix of the heap corruption continued:
# call e showed that neither the direct copies (L1 0/50) nor the round-5 teardown order (L2 0/50) alone bring it back, and this tree is clean (S 0/100).
#   L3   both legacy behaviours together                                                   x50
#   LAZY this tree's objects linked WITHOUT -z now and loaded with lazy binding (RTLD_LAZY)  x50   (the round-5 loader behaviour)
#   LZ3  LAZY + both legacy behaviours = the round-5 library in everything but the kernels  x40
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/gpu_calls/gpu_r6f.sh > gpurun_out/r6f_stdout.txt 2>&1
O=$PWD/gpurun_out/r6g
mkdir -p $O
export TMPDIR=/tmp
K="test_bdpt_full_matches_reference_at_4096_spp and classic"
summary=$O/summary.txt
: > $summary
loop() { # name count legacy lazy
  name=$1; count=$2; legacy=$3; lazy=$4
  crashes=0; fails=0
  for i in $(seq 1 $count); do
    if [ -n "$lazy" ]; then
      ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_lazy.so ETX_HIP_DLOPEN_LAZY=1 ETX_HIP_DEBUG_LEGACY=$legacy timeout 300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/${name}_$i.log 2>&1
    else
      ETX_HIP_DEBUG_LEGACY=$legacy timeout 300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/${name}_$i.log 2>&1
    fi
    rc=$?
    if [ $rc = 139 ] || [ $rc = 134 ]; then crashes=$((crashes + 1)); elif [ $rc != 0 ]; then fails=$((fails + 1)); fi
    if [ $rc = 0 ]; then rm -f $O/${name}_$i.log; else echo "$name $i rc=$rc $(grep -m1 -E 'Error|error' $O/${name}_$i.log | cut -c1-160)" >> $summary; fi
    rm -f core*
  done
  echo "$name: $count runs, $crashes crashed, $fails failed otherwise (ETX_HIP_DEBUG_LEGACY='$legacy' lazy='$lazy')" >> $summary
}
loop LZ3 40 3 1
loop LAZY 50 0 1
loop L3 50 3 ""
cat gpurun_out/r6f_stdout.txt | tail -40
cat $summary
