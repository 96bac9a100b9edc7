// kernels_tail.hip - tail kernels of the wavefront loop.
//
// Russian roulette (scene.hxx:228-248, continuation probability up to 0.95) leaves a long, thin tail: after ~25 bounces
// fewer than 2 % of the paths are alive, but the last one dies after 150-300 bounces. Running the tail as one launch
// per bounce costs ~10 us of dispatch per kernel for a handful of rays, which was a third of the iteration time. Once
// the active count drops below capacity / 64 the host launches ONE of these kernels instead: every lane owns a path and
// loops {closest hit (inline traversal), shade step} until the path dies. The step functions are the ones the
// wavefront kernels use (dev_vcm_steps.h); shadow segments and camera vertices accumulate in the same queues and are
// drained by one k_trace_shadow / k_expand_pairs / k_connect_pairs / k_merge launch after the tail.
#include "kernels_shade.inl"  // k_path_tail template; the general-material instantiations are separate translation units

namespace etxd {

__global__ void k_reset_round_counters(Pipeline p) {
  if ((blockIdx.x == 0) && (threadIdx.x == 0)) {
    if (p.counters[kCntCameraVertices] != 0u)
      atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatCameraVertices), (unsigned long long)p.counters[kCntCameraVertices]);
    p.counters[kCntCameraVertices] = 0u;
    p.counters[kCntPairs] = 0u;
    p.counters[kCntShadow] = 0u;
    p.counters[kCntMergeVertices] = 0u;
    p.counters[kCntEndpoints] = 0u;
  }
}

void launch_light_tail(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups) {
  hipLaunchKernelGGL(k_reset_round_counters, dim3(1), dim3(64), 0, stream, p);
  if (groups.binned() == false)
    hipLaunchKernelGGL((k_path_tail<false, kShadeGroupSimple>), dim3(tail_blocks(max_items)), dim3(kBlockSize), 0, stream, p, it, in_set);
  else
    launch_light_tail_group(stream, p, it, in_set, tail_blocks(max_items), groups.widest());
}

void launch_camera_tail(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups) {
  hipLaunchKernelGGL(k_reset_round_counters, dim3(1), dim3(64), 0, stream, p);
  if (groups.binned() == false)
    hipLaunchKernelGGL((k_path_tail<true, kShadeGroupSimple>), dim3(tail_blocks(max_items)), dim3(kBlockSize), 0, stream, p, it, in_set);
  else
    launch_camera_tail_group(stream, p, it, in_set, tail_blocks(max_items), groups.widest());
}

}  // namespace etxd
