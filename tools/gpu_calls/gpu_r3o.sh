#!/bin/bash
# round 3, GPU call o: occupancy of the opaque shadow kernel (LDS stack 32 / 16 entries, 5-7 waves per SIMD), short-stack closest-hit kernel,
# path table length, tail threshold - interleaved A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3o
mkdir -p $O
export TMPDIR=/tmp
V=$PWD/etx-tracer_amd/variants
run() {  # workload tag [ENV=...]
  w=$1; tag=$2; shift 2
  lib=$V/libetx_hip_$tag.so; [ "$tag" = base ] && lib=$PWD/etx-tracer_amd/libetx_hip.so
  v=$(env "$@" ETX_HIP_LIBRARY=$lib timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "$w $tag $*: $v" >> $O/ab.txt
}
timeout 600 python -m pytest tests/test_gpu_sssmesh.py tests/test_gpu_parity_size.py -x -q -m gpu -k "mesh or config3 or stack" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/log.txt
run sssdragon_bdpt base A=0
run sssdragon_bdpt s32 A=0
run sssdragon_bdpt w6 A=0
run sssdragon_bdpt w7 A=0
run sssdragon_bdpt dbg ETX_HIP_BVH_VARIANT=2
run sssdragon_bdpt dbg ETX_HIP_PATH_TABLE=8
run sssdragon_bdpt base A=0
run sssdragon_bdpt w6 A=0
run sssdragon_bdpt w7 A=0
run gems1m base A=0
run gems1m w6 A=0
run gems1m dbg ETX_HIP_BVH_VARIANT=2
run gems base A=0
run gems dbg ETX_HIP_BVH_VARIANT=2
run full dbg ETX_HIP_TAIL_DIVISOR=64
run full dbg ETX_HIP_TAIL_DIVISOR=32
run full dbg ETX_HIP_TAIL_DIVISOR=16
run full dbg ETX_HIP_TAIL_DIVISOR=64
grep -n "passed\|failed" $O/tests.log | tail -n 3
cat $O/log.txt $O/ab.txt
