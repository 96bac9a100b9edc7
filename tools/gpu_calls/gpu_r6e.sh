#!/bin/bash
# round 6, GPU call e (d ran on a tree whose test collection still imported torch: every run was refused by the new runtime check): which change removes the corruption? Always the driver's pytest command restricted with -k, this tree's library, no torch:
#   S  as built                                        x90
#   L1 ETX_HIP_DEBUG_LEGACY=1: direct copies again     x60
#   L2 ETX_HIP_DEBUG_LEGACY=2: round-5 teardown order  x60
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6e
mkdir -p $O
export TMPDIR=/tmp
K="test_bdpt_full_matches_reference_at_4096_spp and classic"
summary=$O/summary.txt
: > $summary
loop() { # name count legacy
  name=$1; count=$2; legacy=$3
  crashes=0; fails=0
  for i in $(seq 1 $count); do
    ETX_HIP_DEBUG_LEGACY=$legacy timeout 300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/${name}_$i.log 2>&1
    rc=$?
    if [ $rc = 139 ] || [ $rc = 134 ]; then crashes=$((crashes + 1)); elif [ $rc != 0 ]; then fails=$((fails + 1)); fi
    if [ $rc = 0 ]; then rm -f $O/${name}_$i.log; else echo "$name $i rc=$rc $(grep -m1 -E 'Error|error' $O/${name}_$i.log | cut -c1-160)" >> $summary; fi
    rm -f core*
  done
  echo "$name: $count runs, $crashes crashed, $fails failed otherwise (ETX_HIP_DEBUG_LEGACY='$legacy')" >> $summary
}
loop L1 50 1
loop L2 50 2
loop S 70 0
cat $summary
