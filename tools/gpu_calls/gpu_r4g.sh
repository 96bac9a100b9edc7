#!/bin/bash
# Round 4, call g: A/B of the triangles' shading rows read without register caching (call f's build kept a 180-byte Isect in scratch), with the shading
# point expanded before (variant `early`) or after (new) the medium sampling, against the build before them (base).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4g
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for round in 1 2; do
  for v in base new early; do
    lib=$V/libetx_hip_$v.so; [ $v = new ] && lib=etx-tracer_amd/libetx_hip.so
    for w in full classic; do
      r=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload $w --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
      echo "$w $v 4 lanes: $r" >> $O/ab.txt
    done
    r=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full $v 1 lane: $r" >> $O/ab.txt
  done
done
timeout 60 tools/micro/bin/gather_bench 2>&1 | grep -i "records per wave\|no load\|LDS" > $O/gather_small.txt
cat $O/ab.txt; cat $O/gather_small.txt; tail -3 $O/err.txt
