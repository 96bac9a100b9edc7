// kernels_shade.inl - the shade and tail kernel templates, shared by the translation units that instantiate them.
// The simple-material instantiations (<true>) live in kernels_vcm.hip / kernels_tail.hip and compile in about a minute;
// every general-material instantiation (<false>: all eleven BSDF classes, Heitz walks, subsurface walk) is its own
// translation unit (kernels_shade_*_general.hip, kernels_tail_*_general.hip) because each one takes many minutes of
// compiler time - they build in parallel and are not rebuilt when only the simple kernels change.
#pragma once
#include "kernels.h"
#include "dev_vcm_steps.h"

#if !defined(ETX_CAM_ATTR)
#define ETX_CAM_ATTR
#endif
#if !defined(ETX_LIGHT_ATTR)
#define ETX_LIGHT_ATTR
#endif

namespace etxd {

static inline uint32_t shade_grid_for(uint32_t capacity) {
  return min(kPersistentBlocks, (capacity + kBlockSize - 1) / kBlockSize);
}

// vcm_light_step, vcm_shared.hxx:1090-1260 (everything after rt.trace)
template <bool kSimple>
__global__ __launch_bounds__(kBlockSize) ETX_LIGHT_ATTR void k_light_shade(Pipeline p, VcmParams it, uint32_t in_set) {
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const PathSet& out = p.paths[in_set ^ 1u];
  const uint32_t count = p.counters[in_set == 0 ? kCntActiveA : kCntActiveB];
  uint32_t* out_counter = p.counters + (in_set == 0 ? kCntActiveB : kCntActiveA);
  __shared__ BlockScratch s_scratch;
  __shared__ int32_t s_stack[kSimple ? 1 : kStackDepth * kBlockSize];  // inline traversal of the subsurface walk (general materials only)
  const LaneStack stack = {s_stack + (kSimple ? 0u : threadIdx.x), kBlockSize};
  const BlockSlots slots = {&s_scratch};
  ETX_BLOCK_LOOP(count, i) {
    const bool valid = i < count;
    PathState st;
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (valid) {
      st = load_path(in, i);
      h = p.hits[i];
    }
    const bool alive = light_step<kSimple>(p, scene, it, st, h, valid, slots, stack);
    const uint32_t slot = slots.get(alive, out_counter);
    if (alive)
      store_path(out, slot, st);
  }
}

// vcm_camera_step, vcm_shared.hxx:927-1079, without the vertex connections and the merge: connectible vertices are
// written to the camera vertex pool and consumed by k_connect / k_merge of the same bounce.
template <bool kSimple>
__global__ __launch_bounds__(kBlockSize) ETX_CAM_ATTR void k_camera_shade(Pipeline p, VcmParams it, uint32_t in_set) {
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const PathSet& out = p.paths[in_set ^ 1u];
  const uint32_t count = p.counters[in_set == 0 ? kCntActiveA : kCntActiveB];
  uint32_t* out_counter = p.counters + (in_set == 0 ? kCntActiveB : kCntActiveA);
  __shared__ BlockScratch s_scratch;
  __shared__ int32_t s_stack[kSimple ? 1 : kStackDepth * kBlockSize];  // inline traversal of the subsurface walk (general materials only)
  const LaneStack stack = {s_stack + (kSimple ? 0u : threadIdx.x), kBlockSize};
  const BlockSlots slots = {&s_scratch};
  ETX_BLOCK_LOOP(count, i) {
    const bool valid = i < count;
    PathState st;
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (valid) {
      st = load_path(in, i);
      h = p.hits[i];
    }
    const bool alive = camera_step<kSimple>(p, scene, it, st, h, valid, slots, stack);
    const uint32_t slot = slots.get(alive, out_counter);
    if (alive)
      store_path(out, slot, st);
  }
}

// Tail kernel (kernels_tail.hip): every lane owns a path and loops {closest hit (inline traversal), shade step} until
// the path ends.
template <bool kCamera, bool kSimple>
__global__ __launch_bounds__(kBlockSize) void k_path_tail(Pipeline p, VcmParams it, uint32_t in_set) {
  __shared__ int32_t s_stack[kStackDepth * kBlockSize];
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const uint32_t count = p.counters[in_set == 0 ? kCntActiveA : kCntActiveB];
  LaneStack stack = {s_stack + threadIdx.x, kBlockSize};
  unsigned long long rays = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    PathState st = load_path(in, i);
    uint32_t alpha_seed = st.sampler.seed ^ 0x2545f491u;
    bool alive = true;
    while (alive) {
      Hit h = bvh_closest(scene, scene.bvh_nodes, scene.bvh_tris, scene.bvh_root, stack, RayQ{st.ray_o, st.ray_tmin, st.ray_d, st.ray_tmax}, alpha_seed, nullptr);
      rays++;
      const float4 hit = make_float4(h.u, h.v, h.t, __uint_as_float(h.tri));
      alive = kCamera ? camera_step<kSimple>(p, scene, it, st, hit, true, LaneSlots{}, stack) : light_step<kSimple>(p, scene, it, st, hit, true, LaneSlots{}, stack);
    }
  }
  if (rays)
    atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatRaysExtension), rays);
}

static inline uint32_t tail_blocks(uint32_t max_items) {
  return max(1u, min(kPersistentBlocks, (max_items + kBlockSize - 1) / kBlockSize));
}

// general-material instantiations, one translation unit each
void launch_light_shade_general(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, const dim3& grid);
void launch_camera_shade_general(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, const dim3& grid);
void launch_light_tail_general(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t blocks);
void launch_camera_tail_general(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t blocks);

}  // namespace etxd
