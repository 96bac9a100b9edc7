// kernels_shade.inl - the shade and tail kernel templates, shared by the translation units that instantiate them.
//
// Material-binned shading. After the closest-hit query a path belongs to the shading group of its hit material
// (DScene::material_group: simple / general / subsurface; misses and medium events are "simple"). The simple kernel runs
// over the whole active set; in scenes that also hold other groups (kBin) it appends the slots of those paths to the
// group's list instead of shading them, and the group's kernel then runs over that dense list. One rough gem therefore
// costs its own hits, not every Lambert wall hit of the scene, and every kernel carries the register budget of its own
// BSDF classes only. All group kernels append their survivors to the same output set.
// The simple-group instantiations live in kernels_vcm.hip / kernels_tail.hip; the general and subsurface ones are their own
// translation units (kernels_shade_*_general.hip, kernels_tail_*_general.hip) so that everything builds in parallel.
#pragma once
#include "kernels.h"
#include "dev_vcm_steps.h"

namespace etxd {

static inline uint32_t shade_grid_for(uint32_t capacity) {
  return min(kPersistentBlocks, (capacity + kBlockSize - 1) / kBlockSize);
}

ETX_DEV uint32_t hit_shade_group(const DScene& scene, const float4& h) {
  const uint32_t tri = __float_as_uint(h.w);
  return (tri == kInvalid) ? uint32_t(kShadeGroupSimple) : uint32_t(scene.material_group[scene.triangles[tri].material_index]);
}

// Slot of the work item and its hit. Group kernels read the slot from their list.
template <uint32_t kGroup>
ETX_DEV uint32_t shade_item_count(const Pipeline& p, uint32_t in_set) {
  if (kGroup == kShadeGroupSimple)
    return p.counters[in_set == 0 ? kCntActiveA : kCntActiveB];
  return min(p.counters[kGroup == kShadeGroupGeneral ? kCntGroupGeneral : kCntGroupSubsurface], p.capacity);
}

// Simple kernel of a binned scene: hand the paths of the other groups over. Workgroup-uniform (block compaction).
template <class Slots>
ETX_DEV bool bin_foreign_groups(const Pipeline& p, const Slots& slots, uint32_t slot, uint32_t group, bool valid) {
  const bool to_general = valid && (group == kShadeGroupGeneral);
  const uint32_t general_at = slots.get(to_general, p.counters + kCntGroupGeneral);
  if (to_general)
    p.group_list[kShadeGroupGeneral - 1u][general_at] = slot;
  const bool to_subsurface = valid && (group == kShadeGroupSubsurface);
  const uint32_t subsurface_at = slots.get(to_subsurface, p.counters + kCntGroupSubsurface);
  if (to_subsurface)
    p.group_list[kShadeGroupSubsurface - 1u][subsurface_at] = slot;
  return valid && (group == kShadeGroupSimple);
}

// vcm_light_step, vcm_shared.hxx:1090-1260 (everything after rt.trace)
template <uint32_t kGroup, bool kBin>
__global__ __launch_bounds__(kBlockSize) void k_light_shade(Pipeline p, VcmParams it, uint32_t in_set) {
  constexpr bool kWalk = kGroup == kShadeGroupSubsurface;
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const PathSet& out = p.paths[in_set ^ 1u];
  const uint32_t count = shade_item_count<kGroup>(p, in_set);
  uint32_t* out_counter = p.counters + (in_set == 0 ? kCntActiveB : kCntActiveA);
  __shared__ BlockScratch s_scratch;
  __shared__ BlockScratch3 s_scratch3;
  __shared__ int32_t s_stack[kWalk ? kStackDepth * kBlockSize : 1];  // inline traversal of the subsurface walk
  const LaneStack stack = lane_stack(scene, s_stack + (kWalk ? threadIdx.x : 0u), kBlockSize);
  const BlockSlots slots = {&s_scratch, &s_scratch3};
  ETX_BLOCK_LOOP(count, j) {
    bool valid = j < count;
    const uint32_t i = (kGroup == kShadeGroupSimple) ? j : (valid ? p.group_list[kGroup == kShadeGroupSimple ? 0u : kGroup - 1u][j] : 0u);
    PathState st;
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (valid)
      h = p.hits[i];
    if (kBin)
      valid = bin_foreign_groups(p, slots, i, hit_shade_group(scene, h), valid);
    if (valid)
      st = load_path(in, i);
    if (kGroup == kShadeGroupSimple) {  // the step reserves the path's slot together with its vertex and shadow-request slots and stores the path
      (void)light_step<kGroup>(p, scene, it, st, h, valid, slots, stack, &out, out_counter);
    } else {
      const bool alive = light_step<kGroup>(p, scene, it, st, h, valid, slots, stack);
      const uint32_t slot = slots.get(alive, out_counter);
      if (alive)
        store_path(out, slot, st);
    }
  }
}

// vcm_camera_step, vcm_shared.hxx:927-1079, without the vertex connections and the merge: connectible vertices are
// written to the camera vertex pool and consumed by k_connect / k_merge of the same bounce.
template <uint32_t kGroup, bool kBin>
__global__ __launch_bounds__(kBlockSize) void k_camera_shade(Pipeline p, VcmParams it, uint32_t in_set) {
  constexpr bool kWalk = kGroup == kShadeGroupSubsurface;
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const PathSet& out = p.paths[in_set ^ 1u];
  const uint32_t count = shade_item_count<kGroup>(p, in_set);
  uint32_t* out_counter = p.counters + (in_set == 0 ? kCntActiveB : kCntActiveA);
  __shared__ BlockScratch s_scratch;
  __shared__ BlockScratch3 s_scratch3;
  __shared__ int32_t s_stack[kWalk ? kStackDepth * kBlockSize : 1];
  const LaneStack stack = lane_stack(scene, s_stack + (kWalk ? threadIdx.x : 0u), kBlockSize);
  const BlockSlots slots = {&s_scratch, &s_scratch3};
  ETX_BLOCK_LOOP(count, j) {
    bool valid = j < count;
    const uint32_t i = (kGroup == kShadeGroupSimple) ? j : (valid ? p.group_list[kGroup == kShadeGroupSimple ? 0u : kGroup - 1u][j] : 0u);
    PathState st;
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (valid)
      h = p.hits[i];
    if (kBin)
      valid = bin_foreign_groups(p, slots, i, hit_shade_group(scene, h), valid);
    if (valid)
      st = load_path(in, i);
    if (kGroup == kShadeGroupSimple) {
      (void)camera_step<kGroup>(p, scene, it, st, h, valid, slots, stack, &out, out_counter);
    } else {
      const bool alive = camera_step<kGroup>(p, scene, it, st, h, valid, slots, stack);
      const uint32_t slot = slots.get(alive, out_counter);
      if (alive)
        store_path(out, slot, st);
    }
  }
}

// Tail kernel (kernels_tail.hip): every lane owns a path and loops {closest hit (inline traversal), shade step} until
// the path ends. A lane meets any material on its way, so kGroup is the widest group the scene holds.
template <bool kCamera, uint32_t kGroup>
__global__ __launch_bounds__(kBlockSize) void k_path_tail(Pipeline p, VcmParams it, uint32_t in_set) {
  __shared__ int32_t s_stack[kStackDepth * kBlockSize];
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const uint32_t count = p.counters[in_set == 0 ? kCntActiveA : kCntActiveB];
  LaneStack stack = lane_stack(p.scene, s_stack + threadIdx.x, kBlockSize);
  unsigned long long rays = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    PathState st = load_path(in, i);
    uint32_t alpha_seed = st.sampler.seed ^ 0x2545f491u;
    bool alive = true;
    while (alive) {
      Hit h = bvh_closest(scene, global_nodes(scene), scene.bvh_tris, scene.bvh_root, stack, RayQ{st.ray_o, st.ray_tmin, st.ray_d, st.ray_tmax}, alpha_seed, nullptr);
      rays++;
      const float4 hit = make_float4(h.u, h.v, h.t, __uint_as_float(h.tri));
      alive = kCamera ? camera_step<kGroup>(p, scene, it, st, hit, true, LaneSlots{}, stack) : light_step<kGroup>(p, scene, it, st, hit, true, LaneSlots{}, stack);
    }
  }
  if (rays)
    atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatRaysExtension), rays);
}

static inline uint32_t tail_blocks(uint32_t max_items) {
  return max(1u, min(kPersistentBlocks, (max_items + kBlockSize - 1) / kBlockSize));
}

// general / subsurface instantiations, one translation unit per kernel family
void launch_light_shade_group(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, const dim3& grid, uint32_t group);
void launch_camera_shade_group(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, const dim3& grid, uint32_t group);
void launch_light_tail_group(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t blocks, uint32_t group);
void launch_camera_tail_group(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t blocks, uint32_t group);

}  // namespace etxd
