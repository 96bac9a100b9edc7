#!/bin/bash
# Builds libetx_hip.so with its HOST objects (host_api.cpp, host_scene.cpp, host_comm.cpp) under a sanitizer: the device code objects are the
# product's (obj/*.o of csrc/build.sh, untouched - there is no device-side sanitizer on this pool), the host side is compiled host-only.
#   tools/build_sanitized.sh address   -> etx-tracer_amd/variants/libetx_hip_asan.so   (run with LD_PRELOAD=$(tools/build_sanitized.sh --runtime address))
#   tools/build_sanitized.sh thread    -> etx-tracer_amd/variants/libetx_hip_tsan.so
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
RT_DIR=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
if [ "$1" = "--runtime" ]; then
  [ "$2" = thread ] && echo "$RT_DIR/libclang_rt.tsan-x86_64.so" || echo "$RT_DIR/libclang_rt.asan-x86_64.so"
  exit 0
fi
KIND=${1:-address}
SHORT=asan; [ "$KIND" = thread ] && SHORT=tsan
SRC="$HERE/etx-tracer_amd/csrc"
OBJ="$SRC/obj"
[ -f "$OBJ/kernels_vcm.o" ] || bash "$SRC/build.sh"
OUT="$HERE/etx-tracer_amd/variants"
WORK="$OBJ/$SHORT"
mkdir -p "$OUT" "$WORK"
for src in host_api.cpp host_scene.cpp host_comm.cpp host_transfer.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -x hip --offload-host-only -fsanitize=$KIND -fno-omit-frame-pointer -c "$SRC/$src" -o "$WORK/${src%.*}.o" &
done
wait
KERNELS=$(ls "$OBJ"/kernels_*.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$KIND -shared-libsan -o "$OUT/libetx_hip_$SHORT.so" "$WORK"/*.o $KERNELS -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$RT_DIR"
echo "built $OUT/libetx_hip_$SHORT.so"
