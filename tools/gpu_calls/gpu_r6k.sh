#!/bin/bash
# round 6, GPU call k: (1) the driver's GPU suite command, whole, on the shipped library (first full run of this round; includes the new textured-subsurface BDPT test),
# (2) smoke(), (3) the host objects under AddressSanitizer (tools/build_sanitized.sh; device code objects are the product's) on the test the round-5 suite died in, x3,
# and on the multi-context test.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6k
mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/tests_full.log 2>&1
echo "driver's suite rc=$? $(($(date +%s) - t0)) s: $(grep -E 'passed|failed|error' $O/tests_full.log | tail -1)" >> $O/log.txt
timeout 300 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1
echo "smoke rc=$? $(grep smoke: $O/smoke.log | tail -1)" >> $O/log.txt
RT=$(bash tools/build_sanitized.sh --runtime address)
for i in 1 2 3; do
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:log_path=$O/asan_report ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_asan.so \
    timeout 400 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "test_bdpt_full_matches_reference_at_4096_spp and classic" > $O/asan_bdpt_$i.log 2>&1
  echo "asan bdpt classic $i rc=$? $(grep -E 'passed|failed|error' $O/asan_bdpt_$i.log | tail -1)" >> $O/log.txt
done
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:log_path=$O/asan_report ETX_HIP_LIBRARY=$PWD/etx-tracer_amd/variants/libetx_hip_asan.so \
  timeout 600 python3 -m pytest tests/test_gpu_contexts.py tests/test_gpu_checkpoint.py tests/test_gpu_scene_update.py -x -q -m gpu -p no:cacheprovider > $O/asan_contexts.log 2>&1
echo "asan contexts + checkpoint + scene update rc=$? $(grep -E 'passed|failed|error' $O/asan_contexts.log | tail -1)" >> $O/log.txt
ls $O/asan_report* 2>/dev/null | wc -l | xargs echo "asan report files:" >> $O/log.txt
for f in $O/asan_report*; do [ -f "$f" ] && head -40 "$f" >> $O/asan_reports_head.txt; done
cat $O/log.txt
tail -5 $O/tests_full.log
