// kernels_trace.hip - the traversal kernel: ray queue -> hit queue.
//
// Roofline: HBM. Algorithmic traffic per ray = 32 B ray record in + 16 B hit record out = 48 B (the queues are
// index-aligned with the path state, so no per-ray path index is moved); BVH bytes are not counted while the tree is
// L2/LDS resident (Cornell: 3 KB). One lane = one ray; 256-thread blocks, persistent grid (256 CUs x 8 blocks),
// wave-uniform grid-stride loop; per-lane traversal stack in LDS laid out [level][lane] (bank conflict free).
#include "kernels.h"
#include "dev_bvh.h"
#include "dev_vcm.h"
#include "pipeline.h"

namespace etxd {

template <bool kFromCounter, bool kFlat>
__global__ __launch_bounds__(kBlockSize) void k_trace_closest(const DScene scene_arg, const float4* __restrict__ ray_o_tmin, const float4* __restrict__ ray_d_tmax,
  float4* __restrict__ hits, uint32_t* __restrict__ counters, uint32_t active_counter, uint32_t fixed_count) {
  // the flat sweep needs no stack: without the 32 KB of LDS the kernel runs 8 waves per SIMD instead of 5
  __shared__ int32_t s_stack[kFlat ? 1 : kStackDepth * kBlockSize];
  const DScene& scene = scene_arg;  // by value: kernarg (scalar) loads, table pointers known to be global
  const uint32_t count = kFromCounter ? counters[active_counter] : fixed_count;
  if (kFromCounter && (blockIdx.x == 0) && (threadIdx.x == 0)) {
    // housekeeping for the shade kernel that follows: its output counter and the camera vertex pool start empty
    counters[kCntActiveA + kCntActiveB - active_counter] = 0u;
    counters[kCntCameraVertices] = 0u;
    counters[kCntPairs] = 0u;
    counters[kCntShadow] = 0u;
    counters[kCntMergeVertices] = 0u;
    atomicAdd(reinterpret_cast<unsigned long long*>(counters + kStatRaysExtension), (unsigned long long)count);
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stride = gridDim.x * blockDim.x;
  LaneStack stack = {s_stack + (kFlat ? 0u : threadIdx.x), kBlockSize};
  for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x - lane; base < count; base += stride) {
    const uint32_t i = base + lane;
    if (i >= count)
      continue;
    const float4 a = ray_o_tmin[i];
    const float4 b = ray_d_tmax[i];
    uint32_t alpha_seed = __float_as_uint(a.x) ^ (__float_as_uint(b.y) * 0x9e3779b9u) ^ i;
    const RayQ ray = {{a.x, a.y, a.z}, a.w, {b.x, b.y, b.z}, b.w};
    Hit h = kFlat ? bvh_flat_closest(scene, scene.bvh_tris, ray, alpha_seed, nullptr)
                  : bvh_closest(scene, scene.bvh_nodes, scene.bvh_tris, scene.bvh_root, stack, ray, alpha_seed, nullptr);
    hits[i] = make_float4(h.u, h.v, h.t, __uint_as_float(h.tri));
  }
}

void launch_trace_closest(hipStream_t stream, const Pipeline& p, uint32_t set, uint32_t active_counter, uint32_t max_items, bool flat) {
  uint32_t blocks = max(1u, min(kPersistentBlocks, (min(p.capacity, max_items) + kBlockSize - 1) / kBlockSize));
  if (flat)
    hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(blocks), dim3(kBlockSize), 0, stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits, p.counters, active_counter, 0u);
  else
    hipLaunchKernelGGL((k_trace_closest<true, false>), dim3(blocks), dim3(kBlockSize), 0, stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits, p.counters, active_counter, 0u);
}

// ---------------------------------------------------------------------------------------------------------------
// Shadow kernel: segment queue -> transmittance -> film atomics (Raytracing::trace_transmittance, rt.cxx:468-579, plus
// the accumulation the callers do: vcm_cpu.cxx:148-153 light splats, vcm_shared.hxx:1049-1053 camera gathers).
// Algorithmic traffic: 48 B request in, 12 B of float atomics out for visible segments.
template <bool kFlat>
__global__ __launch_bounds__(kBlockSize) void k_trace_shadow(Pipeline p) {
  __shared__ int32_t s_stack[kFlat ? 1 : kStackDepth * kBlockSize];
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntShadow], p.shadow.capacity);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stride = gridDim.x * blockDim.x;
  LaneStack stack = {s_stack + (kFlat ? 0u : threadIdx.x), kBlockSize};  // a flat scene only touches it when a segment crosses > 4 boundaries
  uint32_t splats = 0;
  for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x - lane; base < count; base += stride) {
    const uint32_t i = base + lane;
    f3 value = mk3(0.0f);
    uint32_t target = 0xffffffffu - lane;  // idle lanes: a target nobody shares
    if (i < count) {
      const float4 a = p.shadow.p0_medium[i];
      const float4 b = p.shadow.p1_target[i];
      uint32_t alpha_seed = __float_as_uint(a.x) ^ (__float_as_uint(b.y) * 0x9e3779b9u) ^ i;
      f3 tr = mk3(1.0f);
      if ((p.debug_flags & 4u) == 0u)
        tr = bvh_transmittance(scene, scene.bvh_nodes, scene.bvh_tris, scene.bvh_root, stack, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, __float_as_uint(a.w), p.shadow.value[i].w, alpha_seed);
      target = __float_as_uint(b.w);
      if ((tr.x > kEpsilon) || (tr.y > kEpsilon) || (tr.z > kEpsilon)) {  // SpectralResponse::is_zero
        const float4 v = p.shadow.value[i];
        value = tr * f3{v.x, v.y, v.z};
        if (target & kShadowTargetLight) {
          // vcm_shared.hxx:1229 + vcm_cpu.cxx:148-153 + film.cxx:148: thresholds of the light splat
          if ((max_component(value) <= kEpsilon) || (dot(value, value) <= kEpsilon))
            value = mk3(0.0f);
          else
            splats++;
        }
      }
    }
    if (p.debug_flags & 1u)
      continue;
    // The film atomics are the expensive part of this kernel (measured: 5 of 9 ms). The K connections of one camera
    // vertex sit next to each other in the queue and hit the same pixel, so runs of equal targets are summed across
    // lanes first (segmented inclusive scan) and only the last lane of a run touches the film.
    const uint32_t previous_target = __shfl_up(target, 1);
    const unsigned long long heads = __ballot((lane == 0u) || (previous_target != target));
    const uint32_t run_start = 63u - uint32_t(__clzll(heads & (~0ull >> (63u - lane))));
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const float ox = __shfl_up(value.x, d), oy = __shfl_up(value.y, d), oz = __shfl_up(value.z, d);
      if (lane >= run_start + d)
        value.x += ox, value.y += oy, value.z += oz;
    }
    const bool run_end = (lane == 63u) || (((heads >> (lane + 1u)) & 1ull) != 0ull);
    if (run_end && ((value.x != 0.0f) || (value.y != 0.0f) || (value.z != 0.0f))) {
      float4* dst = (target & kShadowTargetLight) ? (p.light_sum + (target & ~kShadowTargetLight)) : (p.camera_sum + target);
      atomicAdd(&dst->x, value.x), atomicAdd(&dst->y, value.y), atomicAdd(&dst->z, value.z);
    }
  }
  if ((blockIdx.x == 0) && (threadIdx.x == 0))
    atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatRaysShadow), (unsigned long long)count);
  __shared__ unsigned long long s_stat;
  block_stat_add(p, kBlockStatSplats, splats, &s_stat);
}

void launch_trace_shadow(hipStream_t stream, const Pipeline& p, uint32_t max_items, bool flat) {
  uint32_t blocks = max(1u, min(kPersistentBlocks, (min(p.shadow.capacity, max_items) + kBlockSize - 1) / kBlockSize));
  if (flat)
    hipLaunchKernelGGL(k_trace_shadow<true>, dim3(blocks), dim3(kBlockSize), 0, stream, p);
  else
    hipLaunchKernelGGL(k_trace_shadow<false>, dim3(blocks), dim3(kBlockSize), 0, stream, p);
}

void launch_trace_rays(hipStream_t stream, const DScene& scene, const float4* ray_o_tmin, const float4* ray_d_tmax, float4* hits, uint32_t count, bool flat) {
  uint32_t blocks = min(kPersistentBlocks, (count + kBlockSize - 1) / kBlockSize);
  if (blocks == 0)
    return;
  if (flat)
    hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(blocks), dim3(kBlockSize), 0, stream, scene, ray_o_tmin, ray_d_tmax, hits, nullptr, 0u, count);
  else
    hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(blocks), dim3(kBlockSize), 0, stream, scene, ray_o_tmin, ray_d_tmax, hits, nullptr, 0u, count);
}

}  // namespace etxd
