#!/bin/bash
# Builds libetx_hip.so for gfx950 (the only target). In-tree output: etx-tracer_amd/libetx_hip.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libetx_hip.so"
OBJ="$HERE/obj"
mkdir -p "$OBJ"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-variable"
pids=()
for src in kernels_shade_camera_general.hip kernels_tail_camera_general.hip kernels_shade_light_general.hip kernels_tail_light_general.hip kernels_connect.hip kernels_bdpt.hip kernels_pt.hip kernels_trace.hip kernels_vcm.hip kernels_tail.hip kernels_grid.hip host_scene.cpp host_api.cpp host_comm.cpp; do
  obj="$OBJ/${src%.*}.o"
  newest=$(ls -t "$HERE"/*.h "$HERE"/*.inl "$HERE/$src" "$HERE/../../include"/*.h "$HERE/build.sh" | head -1)
  if [ ! -f "$obj" ] || [ "$newest" -nt "$obj" ]; then
    echo "  HIPCC $src"
    ( $HIPCC $FLAGS -x hip -c "$HERE/$src" -o "$obj" ${ETX_HIP_EXTRA_FLAGS} ) & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT"
