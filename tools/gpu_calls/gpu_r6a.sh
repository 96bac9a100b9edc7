#!/bin/bash
# round 6, GPU call a: reproduce the interpreter crash of GPUTEST_r05 (rc 139 in the first test of the suite) and get a NATIVE backtrace.
#   A  driver's command restricted with -k (collects all of tests/: torch and its bundled ROCm 7.0.2 runtime are in the process)
#   G  the same under rocgdb -batch (backtrace of all threads at the SIGSEGV)
#   C  the test file alone (torch never imported: /opt/rocm 7.2 runtime)
#   B  A under MALLOC_CHECK_=3 MALLOC_PERTURB_=165
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6a
mkdir -p $O
export TMPDIR=/tmp
ulimit -c unlimited
{
  echo "core_pattern: $(cat /proc/sys/kernel/core_pattern)"; echo "ulimit -c: $(ulimit -c)"; nproc; free -g | head -2
  python3 - <<'PY'
import ctypes, os
lib = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
v = ctypes.c_int(0); lib.hipRuntimeGetVersion(ctypes.byref(v)); print("/opt/rocm runtime version", v.value)
PY
  python3 - <<'PY'
import torch, ctypes
print("torch", torch.__version__, "hip", torch.version.hip)
lib = ctypes.CDLL("libamdhip64.so.7")
v = ctypes.c_int(0); lib.hipRuntimeGetVersion(ctypes.byref(v)); print("runtime version in a torch process", v.value)
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][:4])
PY
} > $O/env.txt 2>&1
K1="test_bdpt_full_matches_reference_at_4096_spp and classic"
K4="test_bdpt_full_matches_reference_at_4096_spp"
run() { # name, index, extra env..., then the command via "$@"
  :
}
summary=$O/summary.txt
: > $summary
for i in $(seq 1 14); do
  K="$K1"; [ $((i % 4)) = 0 ] && K="$K4"
  t0=$(date +%s.%N)
  timeout 300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/A_$i.log 2>&1
  rc=$?
  echo "A $i rc=$rc $(echo "$(date +%s.%N) - $t0" | bc) s k='$K' $(tail -1 $O/A_$i.log | cut -c1-100)" >> $summary
  if [ $rc != 0 ]; then
    ls -la core* /tmp/core* 2>/dev/null >> $summary
    c=$(ls -t core* 2>/dev/null | head -1)
    if [ -n "$c" ]; then
      timeout 300 /opt/rocm/bin/rocgdb -batch -nx -ex "set pagination off" -ex "thread apply all bt 40" -ex "info sharedlibrary" python3 "$c" > $O/A_${i}_core_bt.txt 2>&1
      rm -f "$c"
    fi
  else
    rm -f $O/A_$i.log
  fi
done
for i in $(seq 1 12); do
  K="$K1"; [ $((i % 4)) = 0 ] && K="$K4"
  t0=$(date +%s.%N)
  timeout 400 /opt/rocm/bin/rocgdb -batch -nx -ex "set pagination off" -ex "set confirm off" -ex "handle SIGSEGV stop print" -ex "handle SIGUSR1 nostop noprint pass" -ex run \
     -ex "echo \n=== STOPPED ===\n" -ex "info registers rip rsp" -ex "x/6i \$pc" -ex "thread apply all bt 40" -ex "info sharedlibrary" -ex "kill" \
     --args python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/G_$i.log 2>&1
  rc=$?
  hit=$(grep -c "SIGSEGV\|SIGABRT\|SIGBUS" $O/G_$i.log)
  echo "G $i rc=$rc signals=$hit $(echo "$(date +%s.%N) - $t0" | bc) s k='$K' $(grep -E "passed|failed" $O/G_$i.log | tail -1 | cut -c1-100)" >> $summary
  [ "$hit" = 0 ] && [ $i != 1 ] && rm -f $O/G_$i.log
done
for i in $(seq 1 10); do
  K="$K1"; [ $((i % 4)) = 0 ] && K="$K4"
  t0=$(date +%s.%N)
  timeout 300 python3 -m pytest tests/test_gpu_bdpt.py -x -q -m gpu -p no:cacheprovider -k "$K" > $O/C_$i.log 2>&1
  rc=$?
  echo "C $i rc=$rc $(echo "$(date +%s.%N) - $t0" | bc) s k='$K' $(tail -1 $O/C_$i.log | cut -c1-100)" >> $summary
  [ $rc = 0 ] && rm -f $O/C_$i.log
  rm -f core*
done
for i in $(seq 1 8); do
  K="$K1"; [ $((i % 4)) = 0 ] && K="$K4"
  t0=$(date +%s.%N)
  MALLOC_CHECK_=3 MALLOC_PERTURB_=165 timeout 300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > $O/B_$i.log 2>&1
  rc=$?
  echo "B $i rc=$rc $(echo "$(date +%s.%N) - $t0" | bc) s k='$K' $(tail -1 $O/B_$i.log | cut -c1-100)" >> $summary
  [ $rc = 0 ] && rm -f $O/B_$i.log
  rm -f core*
done
cat $O/env.txt $summary
