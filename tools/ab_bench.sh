#!/bin/bash
# A/B in ONE gpurun call (boxes differ by a few percent): tools/ab_bench.sh <workload> <repeats> tag[:FLAGS] ...
# (ETX_HIP_DEBUG_FLAGS and the other knobs are read by ETX_HIP_DEBUG builds only: build the variants with tools/build_variant.sh <tag> "" host_api.cpp)
# Every run is bench.py's default three timed regions: the median and the spread are printed.
# tag = base (regular library) or a tools/build_variant.sh tag; FLAGS = ETX_HIP_DEBUG_FLAGS for that run. Interleaved repeats.
w=$1; n=$2; shift 2
for r in $(seq $n); do
  for spec in "$@"; do
    tag=${spec%%:*}; flags=0; [[ "$spec" == *:* ]] && flags=${spec##*:}
    lib=$PWD/etx-tracer_amd/variants/libetx_hip_$tag.so; [ "$tag" = base ] && lib=$PWD/etx-tracer_amd/libetx_hip.so
    v=$(ETX_HIP_DEBUG_FLAGS=$flags ETX_HIP_LIBRARY=$lib python bench.py --workload $w --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], '(regions', d['repeats']['min'], '..', d['repeats']['max'], ')')")
    echo "$w $spec run $r: $v"
  done
done
