#!/bin/bash
# round 6, GPU call j. Call h settled WHERE the heap corruption lives: only with the ROCm 7.0.2 runtime of the torch wheel under the library (round-5 library on
# /opt/rocm 7.2: 0 of 60; this round's code with the three round-5 host behaviours back on, on 7.0.2: 10 of 60; as shipped, on 7.0.2: 0 of 70). Which of the
# three behaviours does the old runtime not survive? One at a time, torch preloaded + ETX_HIP_ALLOW_OLDER_RUNTIME=1:
#   NT1  direct hipMemcpyAsync of the caller's pageable memory (ETX_HIP_DEBUG_LEGACY=1)      x40
#   NT2  round-5 teardown order (ETX_HIP_DEBUG_LEGACY=2)                                       x40
#   NTL  lazy binding (library linked without -z now, dlopen without RTLD_NOW)                 x40
# Before that: call i (the path-row layout: parity subset + A/B).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/gpu_calls/gpu_r6i.sh > gpurun_out/r6i_stdout.txt 2>&1
O=$PWD/gpurun_out/r6j
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
# the lazy variant must hold THIS tree's objects (call i changed the kernels): link it here
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libetx_hip_lazy_now.so etx-tracer_amd/csrc/obj/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>> $summary
( export ETX_TESTS_PRELOAD_TORCH=1 ETX_HIP_ALLOW_OLDER_RUNTIME=1 ETX_HIP_DEBUG_LEGACY=1; loop NT1 40 )
( export ETX_TESTS_PRELOAD_TORCH=1 ETX_HIP_ALLOW_OLDER_RUNTIME=1 ETX_HIP_DEBUG_LEGACY=2; loop NT2 40 )
( export ETX_TESTS_PRELOAD_TORCH=1 ETX_HIP_ALLOW_OLDER_RUNTIME=1 ETX_HIP_LIBRARY=$O/libetx_hip_lazy_now.so ETX_HIP_DLOPEN_LAZY=1; loop NTL 40 )
rm -f $O/libetx_hip_lazy_now.so
cat gpurun_out/r6i_stdout.txt | tail -30
cat $summary
