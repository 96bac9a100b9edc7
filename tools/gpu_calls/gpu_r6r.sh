#!/bin/bash
# round 6, GPU call r: the shipped (final) library under the reproduction loop of calls a-j: the driver's pytest command restricted to the test the round-5 suite died in,
# fresh processes; F = as shipped (ROCm 7.2 runtime, no torch) x40; FT = torch preloaded (ROCm 7.0.2 runtime under the library, ETX_HIP_ALLOW_OLDER_RUNTIME=1) x40;
# then tools/context_stress.py (create / render / destroy cycles of all three integrators on two host threads).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6r
mkdir -p $O
export TMPDIR=/tmp
K="test_bdpt_full_matches_reference_at_4096_spp and classic"
summary=$O/summary.txt
: > $summary
loop() { # name count ; environment comes from the caller
  name=$1; count=$2
  crashes=0; fails=0
  for i in $(seq 1 $count); do
    timeout 300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/${name}_$i.log 2>&1
    rc=$?
    if [ $rc = 139 ] || [ $rc = 134 ]; then crashes=$((crashes + 1)); elif [ $rc != 0 ]; then fails=$((fails + 1)); fi
    if [ $rc = 0 ]; then rm -f $O/${name}_$i.log; else echo "$name $i rc=$rc $(grep -m1 -E 'Fatal|Error|error' $O/${name}_$i.log | cut -c1-160)" >> $summary; fi
    rm -f core*
  done
  echo "$name: $count runs, $crashes crashed, $fails failed otherwise" >> $summary
}
( loop F 40 )
( export ETX_TESTS_PRELOAD_TORCH=1 ETX_HIP_ALLOW_OLDER_RUNTIME=1; loop FT 40 )
timeout 400 python3 tools/context_stress.py > $O/context_stress.txt 2>&1
echo "context_stress rc=$? $(tail -1 $O/context_stress.txt | cut -c1-200)" >> $summary
cat $summary
