// kernels_tail.hip - tail kernels of the wavefront loop.
//
// Russian roulette (scene.hxx:228-248, continuation probability up to 0.95) leaves a long, thin tail: after ~25 bounces
// fewer than 2 % of the paths are alive, but the last one dies after 150-300 bounces. Running the tail as one launch
// per bounce costs ~10 us of dispatch per kernel for a handful of rays, which was a third of the iteration time. Once
// the active count drops below capacity / 64 the host launches ONE of these kernels instead: every lane owns a path and
// loops {closest hit (inline traversal), shade step} until the path dies. The step functions are the ones the
// wavefront kernels use (dev_vcm_steps.h); shadow segments and camera vertices accumulate in the same queues and are
// drained by one k_trace_shadow / k_expand_pairs / k_connect_pairs / k_merge launch after the tail.
#include "kernels.h"
#include "dev_vcm_steps.h"

namespace etxd {

__global__ void k_reset_round_counters(Pipeline p) {
  if ((blockIdx.x == 0) && (threadIdx.x == 0)) {
    p.counters[kCntCameraVertices] = 0u;
    p.counters[kCntPairs] = 0u;
    p.counters[kCntShadow] = 0u;
    p.counters[kCntMergeVertices] = 0u;
  }
}

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

static uint32_t tail_blocks(uint32_t max_items) {
  return max(1u, min(kPersistentBlocks, (max_items + kBlockSize - 1) / kBlockSize));
}

void launch_light_tail(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, bool simple_materials) {
  hipLaunchKernelGGL(k_reset_round_counters, dim3(1), dim3(64), 0, stream, p);
  if (simple_materials)
    hipLaunchKernelGGL((k_path_tail<false, true>), dim3(tail_blocks(max_items)), dim3(kBlockSize), 0, stream, p, it, in_set);
  else
    hipLaunchKernelGGL((k_path_tail<false, false>), dim3(tail_blocks(max_items)), dim3(kBlockSize), 0, stream, p, it, in_set);
}

void launch_camera_tail(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, bool simple_materials) {
  hipLaunchKernelGGL(k_reset_round_counters, dim3(1), dim3(64), 0, stream, p);
  if (simple_materials)
    hipLaunchKernelGGL((k_path_tail<true, true>), dim3(tail_blocks(max_items)), dim3(kBlockSize), 0, stream, p, it, in_set);
  else
    hipLaunchKernelGGL((k_path_tail<true, false>), dim3(tail_blocks(max_items)), dim3(kBlockSize), 0, stream, p, it, in_set);
}

}  // namespace etxd
