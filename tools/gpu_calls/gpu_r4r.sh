#!/bin/bash
# Round 4, call r: A/B of the small tables staged in LDS by the simple group's shade kernels (base = the same build with -DETX_STAGE_TABLES=0);
# a quick parity run of the staged kernels (the tight 4096-spp VCM tests) first.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${ETX_AB_TAG:-r4r}
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
if [ "${ETX_AB_PARITY:-1}" = 1 ]; then
  timeout 400 python -m pytest tests/test_gpu_parity_hi.py tests/test_gpu_parity.py -q -m gpu -x -k "shared_streams or feature_scenes or (vcm_matches and gems) or vcm_full_cornell" 2>&1 | tail -5 > $O/parity.txt
fi
for round in 1 2 3; do
  for v in base new; do
    lib=$V/libetx_hip_$v.so; [ $v = new ] && lib=etx-tracer_amd/libetx_hip.so
    for w in ${ETX_AB_WORKLOADS:-full classic}; do
      r=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload $w --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
      echo "$w $v 4 lanes: $r" >> $O/ab.txt
    done
    r=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full $v 1 lane: $r" >> $O/ab.txt
  done
done
cat $O/parity.txt 2>/dev/null; cat $O/ab.txt; tail -3 $O/err.txt
