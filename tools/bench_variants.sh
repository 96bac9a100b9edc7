#!/bin/bash
# bench.py over experiment builds (tools/build_variant.sh): tools/bench_variants.sh <out file> <tag> [...]; tag "base" = the regular library
out=$1; shift
: > $out
for tag in "$@"; do
  lib=etx-tracer_amd/variants/libetx_hip_$tag.so
  [ "$tag" = base ] && lib=etx-tracer_amd/libetx_hip.so
  for lanes in 4 1; do
    v=$(ETX_HIP_LANES=$lanes ETX_HIP_LIBRARY=$PWD/$lib python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$tag lanes=$lanes $v" | tee -a $out
  done
done
