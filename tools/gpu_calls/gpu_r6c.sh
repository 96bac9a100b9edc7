#!/bin/bash
# round 6, GPU call c: the crash reproduces under pytest (call a: 4 of 26) and NOT in tools/stray_write_probe.py (call b: 0 of 128). Here, always the
# driver's pytest command restricted with -k:
#   O  round-5 library, torch preloaded (the round-5 configuration)          x30
#   S  this tree's library (pinned transfer slots), torch preloaded           x30
#   Q  round-5 library, torch NOT loaded (/opt/rocm 7.2 runtime)              x30
#   G  as O under rocgdb: registers, the corrupted object, its page, mappings x24
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6c
mkdir -p $O
export TMPDIR=/tmp
OLD=$PWD/etx-tracer_amd/variants/libetx_hip_r5.so
NEW=$PWD/etx-tracer_amd/libetx_hip.so
K="test_bdpt_full_matches_reference_at_4096_spp and classic"
summary=$O/summary.txt
: > $summary
loop() { # name count lib preload
  name=$1; count=$2; lib=$3; preload=$4
  crashes=0; fails=0
  for i in $(seq 1 $count); do
    ETX_HIP_LIBRARY=$lib ETX_TESTS_PRELOAD_TORCH=$preload timeout 300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/${name}_$i.log 2>&1
    rc=$?
    if [ $rc = 139 ] || [ $rc = 134 ]; then crashes=$((crashes + 1)); elif [ $rc != 0 ]; then fails=$((fails + 1)); fi
    if [ $rc = 0 ]; then rm -f $O/${name}_$i.log; else echo "$name $i rc=$rc" >> $summary; fi
    rm -f core*
  done
  echo "$name: $count runs, $crashes crashed, $fails failed otherwise (lib $(basename $lib), torch preloaded: '$preload')" >> $summary
}
loop O 30 $OLD 1
loop S 30 $NEW 1
loop Q 30 $OLD ""
for i in $(seq 1 24); do
  ETX_HIP_LIBRARY=$OLD ETX_TESTS_PRELOAD_TORCH=1 timeout 300 /opt/rocm/bin/rocgdb -batch -nx -ex "set pagination off" -ex "set confirm off" -ex "handle SIGSEGV stop print" -ex run \
     -ex "echo \n=== STOPPED ===\n" -ex "info registers" -ex "x/8i \$pc" -ex "echo \n=== OBJECT rbp ===\n" -ex "x/96gx (\$rbp & ~0xff) - 0x100" \
     -ex "echo \n=== PAGE ===\n" -ex "x/512gx (\$rbp & ~0xfff)" -ex "echo \n=== MAPPINGS ===\n" -ex "info proc mappings" -ex "thread apply all bt 24" -ex "kill" \
     --args python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/G_$i.log 2>&1
  hit=$(grep -c "received signal SIG" $O/G_$i.log)
  echo "G $i signals=$hit" >> $summary
  [ "$hit" = 0 ] && rm -f $O/G_$i.log
done
cat $summary
