#!/bin/bash
# Round 4, call s: small tables staged in LDS by the simple group's shade kernels, four builds interleaved:
#   base = no staging, 2048 workgroups; nostage512 = no staging, 512 workgroups; norows = staged without the triangles' rows, 512; new = everything staged, 512
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4s
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for round in 1 2 3; do
  for v in base nostage512 norows new; do
    lib=$V/libetx_hip_$v.so; [ $v = new ] && lib=etx-tracer_amd/libetx_hip.so
    r=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    r1=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full $v: 4 lanes $r, 1 lane $r1" >> $O/ab.txt
  done
done
cat $O/ab.txt; tail -3 $O/err.txt
