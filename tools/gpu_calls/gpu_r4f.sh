#!/bin/bash
# Round 4, call f: A/B of the triangles' shading rows (dev_scene.h kTriShadeStride) and the deferred make_intersection against the build before
# them (variants/libetx_hip_base.so = HEAD c34b340+), with and without the rows kept in registers; parity of the new build on the fog box and the
# gems box at 4096 spp; the gather microbenchmark's small-table case.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4f
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for round in 1 2; do
  for v in base new nocache; do
    lib=$V/libetx_hip_$v.so; [ $v = new ] && lib=etx-tracer_amd/libetx_hip.so
    for w in full classic; do
      r=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload $w --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
      echo "$w $v 4 lanes: $r" >> $O/ab.txt
    done
    r=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full $v 1 lane: $r" >> $O/ab.txt
  done
done
for v in base new; do
  lib=$V/libetx_hip_$v.so; [ $v = new ] && lib=etx-tracer_amd/libetx_hip.so
  for w in gems sssdragon_bdpt; do
    r=$(ETX_HIP_LIBRARY=$lib timeout 200 python bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "$w $v: $r" >> $O/ab.txt
  done
done
timeout 400 python -m pytest tests/test_gpu_parity_hi.py -q -m gpu -s -k "test_vcm_matches_reference_at_4096_spp and (full or gems)" > $O/parity.log 2>&1
echo "parity rc=$?" >> $O/ab.txt
timeout 60 tools/micro/bin/gather_bench 2>&1 | grep -i "records per wave\|no load\|LDS" > $O/gather_small.txt
cat $O/ab.txt; tail -4 $O/parity.log; cat $O/gather_small.txt; tail -3 $O/err.txt
