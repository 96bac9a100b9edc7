#!/bin/bash
# Builds oracle/_ref/etx_oracle: the reference's own integrator / shading / film / scene-loader sources, compiled
# read-only from /root/reference, plus the shims in oracle/shims (no Embree / OIDN / enkiTS-pimpl / Linux platform).
# Outputs go only to oracle/_ref/ (git-ignored, travels to the GPU box with gpurun).
set -e
REF=${ETX_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
OBJ="$OUT/obj"
mkdir -p "$OBJ"
if [ ! -d "$REF/sources/etx" ]; then
  echo "reference not present at $REF - keeping prebuilt $OUT" >&2
  exit 0
fi
CXX=/opt/rocm/lib/llvm/bin/clang++   # g++ 11 rejects sources/etx/util/options.hxx:78 (in-class explicit specialisation)
CC=/opt/rocm/lib/llvm/bin/clang
FLAGS="-Wno-invalid-offsetof -std=c++23 -O2 -g0 -DNDEBUG -D_stricmp=strcasecmp -DETX_HAVE_OPENVDB=1 -D_USE_MATH_DEFINES=1 -DETX_LIBRARY=1 -march=native -w -fPIC"
T="$REF/thirdparty"
INC="-I$HERE/../include -I$HERE/../integration -I$REF/sources -I$T -I$T/enkits -I$T/bluenoise -I$T/json -I$T/tinyobjloader -I$T/tinygltf -I$T/mikktspace -I$T/stb_image -I$T/tinyexr -I$T/nanovdb"

compile() { # src obj
  if [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ "$0" -nt "$2" ] || [ "$HERE/../integration/etx_hip_integrators.hxx" -nt "$2" -a "$(basename $1)" = "etx_oracle.cxx" ] || [ "$HERE/../include/etx_hip.h" -nt "$2" -a "$(basename $1)" = "etx_oracle.cxx" ]; then
    echo "  CXX $(basename $1)"
    $CXX $FLAGS $INC -c "$1" -o "$2"
  fi
}

pids=()
compile "$HERE/ref/unity_core.cxx"        "$OBJ/unity_core.o" & pids+=($!)
compile "$HERE/ref/unity_render.cxx"      "$OBJ/unity_render.o" & pids+=($!)
compile "$HERE/ref/unity_rt.cxx"          "$OBJ/unity_rt.o" & pids+=($!)
compile "$HERE/shims/platform_linux.cxx"  "$OBJ/platform_linux.o" & pids+=($!)
compile "$HERE/shims/tasks_threads.cxx"   "$OBJ/tasks_threads.o" & pids+=($!)
compile "$HERE/shims/denoiser_stub.cxx"   "$OBJ/denoiser_stub.o" & pids+=($!)
compile "$HERE/shims/raytracing_bvh.cxx"  "$OBJ/raytracing_bvh.o" & pids+=($!)
compile "$HERE/driver/etx_oracle.cxx"     "$OBJ/etx_oracle.o" & pids+=($!)
compile "$HERE/ref/abi_check.cxx"         "$OBJ/abi_check.o" & pids+=($!)
# the headless host of the HIP backend alone: the same driver without the CPU integrators
mkdir -p "$OBJ/hip_only"
compile "$HERE/ref/unity_rt_host.cxx"     "$OBJ/hip_only/unity_rt_host.o" & pids+=($!)
( if [ ! -f "$OBJ/hip_only/etx_hip_render.o" ] || [ "$HERE/driver/etx_oracle.cxx" -nt "$OBJ/hip_only/etx_hip_render.o" ] || [ "$HERE/../integration/etx_hip_integrators.hxx" -nt "$OBJ/hip_only/etx_hip_render.o" ] || [ "$HERE/../include/etx_hip.h" -nt "$OBJ/hip_only/etx_hip_render.o" ] || [ "$0" -nt "$OBJ/hip_only/etx_hip_render.o" ]; then
    echo "  CXX etx_oracle.cxx (hip only)"; $CXX $FLAGS -DETX_DRIVER_HIP_ONLY $INC -c "$HERE/driver/etx_oracle.cxx" -o "$OBJ/hip_only/etx_hip_render.o"; fi ) & pids+=($!)
for f in bluenoise/bluenoise.cxx stb_image/stb_image.cxx tinyexr/tinyexr.cxx tinygltf/tiny_gltf.cxx tinyobjloader/tiny_obj_loader.cxx; do
  compile "$T/$f" "$OBJ/$(basename ${f%.cxx}).o" & pids+=($!)
done
if [ ! -f "$OBJ/mikktspace.o" ]; then $CC -O2 -w -fPIC -c "$T/mikktspace/mikktspace.c" -o "$OBJ/mikktspace.o" & pids+=($!); fi
for p in "${pids[@]}"; do wait $p; done

$CXX -O2 -o "$OUT/etx_oracle" "$OBJ"/*.o -lpthread -ldl
echo "built $OUT/etx_oracle"
HOST_OBJECTS=$(ls "$OBJ"/*.o | grep -v "/unity_rt.o$" | grep -v "/etx_oracle.o$")
$CXX -O2 -o "$OUT/etx_hip_render" $HOST_OBJECTS "$OBJ/hip_only/unity_rt_host.o" "$OBJ/hip_only/etx_hip_render.o" -lpthread -ldl
echo "built $OUT/etx_hip_render"
