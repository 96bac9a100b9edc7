#!/bin/bash
# round 6, GPU call b: tools/stray_write_probe.py on the ROUND-5 library (etx-tracer_amd/variants/libetx_hip_r5.so): who writes into memory the
# process has given back? torch (ROCm 7.0.2 runtime) vs no torch (/opt/rocm 7.2), with / without film read-back, canary mappings, and a full
# register / memory dump at the SIGSEGV under rocgdb.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6b
mkdir -p $O
export TMPDIR=/tmp
export ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_r5.so
summary=$O/summary.txt
: > $summary
loop() { # name count args...
  name=$1; count=$2; shift 2
  crashes=0; strays=0
  for i in $(seq 1 $count); do
    timeout 120 python3 tools/stray_write_probe.py "$@" > $O/${name}_$i.log 2>&1
    rc=$?
    [ $rc = 139 ] || [ $rc = 134 ] && crashes=$((crashes + 1))
    [ $rc = 3 ] && strays=$((strays + 1))
    if [ $rc = 0 ]; then rm -f $O/${name}_$i.log; else echo "$name $i rc=$rc $(grep PROBE $O/${name}_$i.log | cut -c1-400)" >> $summary; fi
  done
  echo "$name: $count runs, $crashes crashed, $strays with stray writes ($*)" >> $summary
}
loop T 30 --torch
loop N 30
loop TF 20 --torch --no-film
loop TV 16 --torch --integrator vcm
loop T1 16 --torch --spp 512
for i in $(seq 1 16); do
  timeout 200 /opt/rocm/bin/rocgdb -batch -nx -ex "set pagination off" -ex "set confirm off" -ex "handle SIGSEGV stop print" -ex run \
     -ex "echo \n=== STOPPED ===\n" -ex "info registers" -ex "x/8i \$pc" -ex "echo \n=== OBJECT rbp ===\n" -ex "x/64gx (\$rbp & ~0xff) - 0x100" \
     -ex "echo \n=== PAGE ===\n" -ex "x/512gx (\$rbp & ~0xfff)" -ex "echo \n=== MAPPINGS ===\n" -ex "info proc mappings" -ex "thread apply all bt 24" -ex "kill" \
     --args python3 tools/stray_write_probe.py --torch --canaries 0 --watch 0 > $O/G_$i.log 2>&1
  hit=$(grep -c "received signal SIG" $O/G_$i.log)
  echo "G $i signals=$hit" >> $summary
  [ "$hit" = 0 ] && rm -f $O/G_$i.log
done
cat $summary
