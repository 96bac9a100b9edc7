#!/bin/bash
# Round 4, call n: A/B of the emitter records read as 16-byte rows in next event estimation (base = the committed build before it).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${ETX_AB_TAG:-r4n}
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for round in 1 2 3; do
  for v in base new; do
    lib=$V/libetx_hip_$v.so; [ $v = new ] && lib=etx-tracer_amd/libetx_hip.so
    for w in full classic; do
      r=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload $w --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
      echo "$w $v 4 lanes: $r" >> $O/ab.txt
    done
    r=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full $v 1 lane: $r" >> $O/ab.txt
  done
done
cat $O/ab.txt; tail -3 $O/err.txt
