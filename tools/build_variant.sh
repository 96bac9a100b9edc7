#!/bin/bash
# Experiment builds: tools/build_variant.sh <tag> "<extra hipcc flags>" <tu.hip> [...]  ->  etx-tracer_amd/variants/libetx_hip_<tag>.so
# Recompiles only the named translation units with the extra flags and links them with the objects of the regular build
# (etx-tracer_amd/csrc/build.sh must have run). Select a variant at run time with ETX_HIP_LIBRARY=<path>.
# The named units are built with -DETX_HIP_DEBUG: they read the ETX_HIP_* tuning knobs (csrc/tuning_knobs.h) the product library ignores -
# name host_api.cpp / host_scene.cpp / kernels_trace.hip to get theirs. Variants are scratch: delete etx-tracer_amd/variants/ after the
# experiment (everything under etx-tracer_amd/ travels to the GPU box and is loaded by path only).
set -e
tag=$1; flags=$2; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC=$ROOT/etx-tracer_amd/csrc
OUT=$ROOT/etx-tracer_amd/variants
TMP=$OUT/obj_$tag
mkdir -p $TMP
cp $CSRC/obj/*.o $TMP/
pids=()
for src in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-unused-variable -DETX_HIP_DEBUG $flags -x hip -c $CSRC/$src -o $TMP/${src%.*}.o ) & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libetx_hip_$tag.so $TMP/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf $TMP
echo "built $OUT/libetx_hip_$tag.so"
