#!/bin/bash
# round 6, GPU call h. Calls c-g never separated the two suspects: every crash so far (9 of 110 pytest processes) had the round-5 library AND the ROCm
# 7.0.2 runtime of the torch wheel in the process (the "no torch" loop Q of call c still imported torch at collection through the anyio plugin's
# getattr on the lazy proxy), and every clean loop of this round's library (0 of 310) ran on /opt/rocm's 7.2 runtime. Here:
#   R5N  round-5 library, torch NOT in the process (ROCm 7.2 runtime)                                                       x60
#   NT3  this round's objects linked lazily, loaded lazily, both round-5 host behaviours back on, torch preloaded (7.0.2)   x60
#   NT0  this round's library as shipped, torch preloaded, ETX_HIP_ALLOW_OLDER_RUNTIME=1                                    x40
# then: tail_divisor on the gems workload (ETX_HIP_DEBUG build of host_api.cpp reads ETX_HIP_TAIL_DIVISOR).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6h
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
python3 -c "
import sys, subprocess
print('torch in a collection process:', subprocess.run([sys.executable, '-c', 'import sys, pytest; pytest.main([\"tests/\", \"-q\", \"-m\", \"gpu\", \"--collect-only\", \"-p\", \"no:cacheprovider\"]); open(\"$O/collect_probe.txt\", \"w\").write(str(\"torch\" in sys.modules))'], capture_output=True).returncode, open('$O/collect_probe.txt').read())" >> $summary 2>&1
( export ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_r5.so; loop R5N 60 )
( export ETX_TESTS_PRELOAD_TORCH=1 ETX_HIP_ALLOW_OLDER_RUNTIME=1 ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_lazy.so ETX_HIP_DLOPEN_LAZY=1 ETX_HIP_DEBUG_LEGACY=3; loop NT3 60 )
( export ETX_TESTS_PRELOAD_TORCH=1 ETX_HIP_ALLOW_OLDER_RUNTIME=1; loop NT0 40 )
for r in 1 2; do
  for div in 32 0 8 16 64 128; do
    x=$(ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_dbg.so ETX_HIP_TAIL_DIVISOR=$div timeout 300 python3 bench.py --workload gems --steps 12 --warmup 4 --repeats 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'])")
    echo "gems tail_divisor $div run $r: $x" >> $O/tail_divisor_gems.txt
  done
done
cat $summary $O/tail_divisor_gems.txt
