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
#include "tuning_knobs.h"

namespace etxd {

// LDS budget of the BVH traversal kernels: the per-lane stack (kStackDepth x 256 lanes x 4 B = 32 KB) plus the staged top
// of the tree (kLdsNodes x 128 B). 256 nodes = 32 KB -> 64 KB per workgroup, two workgroups per CU; a BVH4 over a few
// thousand triangles fits completely (gems: 2 892 triangles), larger trees keep their top five levels in LDS.
constexpr uint32_t kLdsNodes = 256;

// Housekeeping of a wavefront round, done by one thread of the round's traversal kernel for the shade / connect kernels that follow:
// the per-bounce queues start empty, the statistics of the bounce that just ended are folded, and the host's view of the wavefront
// (host_api.cpp run_bounce_loop) gets its entry: (round tag + 1, live paths entering this round) in pinned host memory - the host
// never drains the stream to learn that a pass has ended.
ETX_DEV void round_housekeeping(uint32_t* __restrict__ counters, uint32_t active_counter, uint32_t count, uint32_t pass_stat, unsigned long long* round_mirror, uint32_t round_tag) {
  counters[kCntActiveA + kCntActiveB - active_counter] = 0u;
  if (counters[kCntCameraVertices] != 0u)  // statistics: the connectible camera vertices of the bounce that just ended
    atomicAdd(reinterpret_cast<unsigned long long*>(counters + kStatCameraVertices), (unsigned long long)counters[kCntCameraVertices]);
  counters[kCntCameraVertices] = 0u;
  counters[kCntPairs] = 0u;
  counters[kCntPairsGeneral] = 0u;
  counters[kCntShadow] = 0u;
  counters[kCntMergeVertices] = 0u;
  counters[kCntEndpoints] = 0u;
  counters[kCntGroupGeneral] = 0u;
  counters[kCntLightBounceBegin] = counters[kCntLightVertices];
  counters[kCntGroupSubsurface] = 0u;
  // subsurface walks of the bidirectional integrator (kernels_bdpt.hip): this round works on walk queue `set` (what the last round left
  // unfinished + what this round's shade kernel adds) and fills queue `set ^ 1`; walks in flight count as live paths for the host
  const uint32_t set = (active_counter == kCntActiveA) ? 0u : 1u;
  const uint32_t walking = counters[kCntWalk + 32u * set];
  counters[kCntWalk + 32u * (set ^ 1u)] = 0u;
  counters[kCntWalkFetch] = 0u;
  counters[kCntWalkExit] = 0u;
  atomicAdd(reinterpret_cast<unsigned long long*>(counters + kStatRaysExtension), (unsigned long long)count);
  if (pass_stat != 0u)
    atomicAdd(reinterpret_cast<unsigned long long*>(counters + pass_stat), (unsigned long long)count);
  if (round_mirror != nullptr)
    __hip_atomic_store(round_mirror + (round_tag & (kRoundMirrorSlots - 1u)), ((unsigned long long)(round_tag) + 1ull) << 32u | (unsigned long long)(count + walking), __ATOMIC_RELEASE,
      __HIP_MEMORY_SCOPE_SYSTEM);
}

// kCross (pipeline, flat scenes that hold Boundary materials): a path that is in NO medium and whose closest hit is a medium boundary
// crosses it right here - vcm_handle_boundary_bsdf (vcm_shared.hxx:436-449) draws nothing and leaves only the medium, the ray origin and the
// path distance changed; handle_surface's Boundary branch (bidirectional.cxx:586-593) comes after three next_2d draws of the vertex
// (six numbers: the crossing here advances the path's stream by them, so that under the reference's seeding the two stay aligned draw by draw) -
// and is traced again from the other side; the shade kernel then meets the segment INSIDE the medium. In the fog box that is the primary segment of every camera path and every
// segment that leaves a wall: a quarter of all segments no longer cost a round of their own. A path that is inside a medium when it
// reaches a boundary is left to the shade kernel (the medium is sampled first).
enum : uint32_t { kCrossNone = 0, kCrossVcm = 1, kCrossBdpt = 2 };
template <bool kFromCounter, bool kFlat, bool kCross = false>
__global__ __launch_bounds__(kBlockSize) void k_trace_closest(const DScene scene_arg, const float4* ray_o_tmin, const float4* ray_d_tmax,
  float4* __restrict__ hits, uint32_t* __restrict__ counters, uint32_t active_counter, uint32_t fixed_count, unsigned long long* round_mirror, uint32_t round_tag, uint32_t lds_node_limit, uint32_t pass_stat,
  PathSet set = {}, uint32_t cross_mode = kCrossNone, unsigned long long* block_stats = nullptr) {
  // the flat sweep needs no stack: without the 32 KB of LDS the kernel runs 8 waves per SIMD instead of 5
  __shared__ int32_t s_stack[kFlat ? 1 : kStackDepth * kBlockSize];
  __shared__ float4 s_nodes[kFlat ? 1 : kLdsNodes * 8u];
  const DScene& scene = scene_arg;  // by value: kernarg (scalar) loads, table pointers known to be global
  const uint32_t count = kFromCounter ? counters[active_counter] : fixed_count;
  if (kFromCounter && (blockIdx.x == 0) && (threadIdx.x == 0)) {
    round_housekeeping(counters, active_counter, count, pass_stat, round_mirror, round_tag);
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stride = gridDim.x * blockDim.x;
  LaneStack stack = lane_stack(scene, s_stack + (kFlat ? 0u : threadIdx.x), kBlockSize);
  BvhNodes nodes = global_nodes(scene);
  if ((kFlat == false) && (blockIdx.x * blockDim.x < count))  // workgroup-uniform: this workgroup has rays
    nodes = stage_nodes(scene, s_nodes, min(kLdsNodes, lds_node_limit));
  unsigned long long crossed_queries = 0ull;
  for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x - lane; base < count; base += stride) {
    const uint32_t i = base + lane;
    if (i >= count)
      continue;
    const float4 a = ray_o_tmin[i];
    const float4 b = ray_d_tmax[i];
    uint32_t alpha_seed = __float_as_uint(a.x) ^ (__float_as_uint(b.y) * 0x9e3779b9u) ^ (__float_as_uint(a.z) * 0x85ebca6bu) ^ (__float_as_uint(b.x) * 0xc2b2ae35u);  // of the ray itself, not of its queue slot (run-dependent)
    const RayQ ray = {{a.x, a.y, a.z}, a.w, {b.x, b.y, b.z}, b.w};
    uint32_t flags = 0u;
    Hit h = kFlat ? bvh_flat_closest(scene, scene.bvh_tris, ray, alpha_seed, kCross ? &flags : nullptr)
                  : bvh_closest(scene, nodes, scene.bvh_tris, scene.bvh_root, stack, ray, alpha_seed, kCross ? &flags : nullptr);
    if (kCross) {
      uint32_t crossings = 0u, medium = kInvalid;
      float crossed = 0.0f;
      f3 origin = ray.o;
      while ((h.tri != kInvalid) && (flags & kTriBoundary) && (crossings < 8u)) {
        if (crossings == 0u)
          medium = set.meta[i].z;
        if (medium != kInvalid)
          break;
        const etx_abi_triangle& tri = scene.triangles[h.tri];
        const etx_abi_material& mat = scene.materials[tri.material_index];
        medium = (dot(ld3(tri.geo_n), ray.d) < 0.0f) ? mat.int_medium : mat.ext_medium;
        crossed += h.t;
        origin = shading_pos(scene, tri, barycentrics(h.u, h.v), ray.d);
        crossings += 1u;
        const RayQ beyond = {origin, kRayEpsilon, ray.d, kMaxFloat};
        h = kFlat ? bvh_flat_closest(scene, scene.bvh_tris, beyond, alpha_seed, &flags)
                  : bvh_closest(scene, nodes, scene.bvh_tris, scene.bvh_root, stack, beyond, alpha_seed, &flags);
      }
      crossed_queries += crossings;
      if (crossings != 0u) {
        const_cast<float4*>(ray_o_tmin)[i] = mk4(origin, kRayEpsilon);
        const_cast<float4*>(ray_d_tmax)[i] = mk4(ray.d, kMaxFloat);
        uint4 meta = set.meta[i];
        meta.z = medium;
        if (cross_mode == kCrossBdpt) {  // the six numbers handle_surface draws before its Boundary branch (bidirectional.cxx:586-593): the path's stream stays where the reference's is
          Sampler stream;
          stream.seed = meta.x;
          for (uint32_t k = 0; k < crossings * 6u; ++k)
            (void)stream.next();
          meta.x = stream.seed;
        }
        set.meta[i] = meta;
        if (cross_mode == kCrossVcm) {  // state.path_distance += intersection.t
          float4 mis = set.mis[i];
          mis.w += crossed;
          set.mis[i] = mis;
        }
      }
    }
    hits[i] = make_float4(h.u, h.v, h.t, __uint_as_float(h.tri));
  }
  if (kCross) {  // the queries beyond crossed boundaries are rays too (statistics: one row per workgroup, no atomics)
    __shared__ unsigned long long s_stat;
    Pipeline stats_only = {};
    stats_only.block_stats = block_stats;
    block_stat_add(stats_only, kBlockStatCrossings, crossed_queries, &s_stat);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// BVH traversal kernel with dynamic ray fetch (persistent threads, Aila & Laine): one lane = one ray, but a lane whose
// ray is finished does not wait for the slowest ray of its wavefront. Measured on the gems scene a ray needs 5 node visits
// and 2.5 triangle tests ON AVERAGE, while the rays that cross a gem need ten times that: with a fixed ray per lane a
// wavefront costs its worst lane and 64-lane utilisation was ~10 % (4 Grays/s). Here every wavefront owns a contiguous
// chunk of the queue and whenever kRefillLanes lanes are idle they take the next rays of the chunk (wave-uniform cursor,
// no atomics).
// PHASES (round 5). Rounds 2-4 ran "one traversal step (inner node or leaf) per lane and loop trip": counters of that kernel alone on the gems
// scene (profiles/round5_pmc_trace_alone.txt) showed its VALU pipes issuing 65 % of the time with 36 % of the lanes active per instruction - every
// trip ran the node step for the lanes on inner nodes AND the leaf step for the lanes on leaves, each lane idle through the other's code. Now a
// wavefront alternates between NODE steps, while at least `node_phase_lanes` lanes stand on inner nodes - a lane that reaches a leaf keeps it
// (one postponed leaf per lane, Aila & Laine's speculative traversal: it pops on and keeps walking, testing boxes against a `best.t` that does not
// know the postponed leaf yet: a few extra node visits, never a wrong answer) - and LEAF steps, in which every lane that holds a leaf tests one
// triangle per trip. Lane utilisation 36 -> 50 %, 9.6 -> 12.0 Grays/s on 2 M incoherent gems rays (profiles/round5_trace_kernel_phases.txt).
// LDS per workgroup: the per-lane stacks (<= 32 KB) + the first kLdsNodesPersistent nodes of the tree (top levels, 8 KB).
constexpr uint32_t kRefillLanes = 16;
constexpr uint32_t kNodePhaseLanes = 24;

template <bool kDeep, uint32_t kLdsEntries = kStackDepth>
struct TraversalStack {
  typedef FastLaneStack Type;
  static ETX_DEV Type make(const DScene&, int32_t* lds_slot, uint32_t stride) {
    return {lds_slot, stride};
  }
};
template <>
struct TraversalStack<true, kStackDepth> {  // the tree's bound exceeds the LDS part: entries above kStackDepth go to DScene::stack_spill
  typedef LaneStack Type;
  static ETX_DEV Type make(const DScene& scene, int32_t* lds_slot, uint32_t stride) {
    return lane_stack(scene, lds_slot, stride);
  }
};
template <>
struct TraversalStack<true, kShortStackDepth> {  // half the LDS: entries above kShortStackDepth spill
  typedef ShortLaneStack Type;
  static ETX_DEV Type make(const DScene& scene, int32_t* lds_slot, uint32_t stride) {
    return short_lane_stack(scene, lds_slot, stride);
  }
};

template <bool kFromCounter, uint32_t kStack, uint32_t kLdsNodesPersistent, bool kDeep = false>
__global__ __launch_bounds__(kBlockSize) void k_trace_closest_bvh(const DScene scene_arg, const float4* __restrict__ ray_o_tmin, const float4* __restrict__ ray_d_tmax,
  float4* __restrict__ hits, uint32_t* __restrict__ counters, uint32_t active_counter, uint32_t fixed_count, unsigned long long* round_mirror, uint32_t round_tag, uint32_t lds_node_limit,
  uint32_t refill_lanes, uint32_t node_phase_lanes, uint32_t pass_stat) {
  __shared__ int32_t s_stack[kStack * kBlockSize];
  __shared__ float4 s_nodes[kLdsNodesPersistent * 8u];
  const DScene& scene = scene_arg;
  const uint32_t count = kFromCounter ? counters[active_counter] : fixed_count;
  if (kFromCounter && (blockIdx.x == 0) && (threadIdx.x == 0)) {
    round_housekeeping(counters, active_counter, count, pass_stat, round_mirror, round_tag);
  }
  if (count == 0u)
    return;
  const BvhNodes nodes = stage_nodes(scene, s_nodes, min(kLdsNodesPersistent, lds_node_limit));
  const typename TraversalStack<kDeep, kDeep ? kStack : kStackDepth>::Type stack = TraversalStack<kDeep, kDeep ? kStack : kStackDepth>::make(scene, s_stack + threadIdx.x, kBlockSize);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6u;
  const uint32_t wave_count = (gridDim.x * blockDim.x) >> 6u;
  const uint32_t chunk = ((count + wave_count - 1u) / wave_count + 63u) & ~63u;  // rays per wavefront, whole 64-ray rows
  uint32_t cursor = min(count, wave * chunk);                                    // wave-uniform
  const uint32_t chunk_end = min(count, cursor + chunk);
  const int32_t kDone = kBvhEmptyChild;
  const BvhTri* __restrict__ tris = scene.bvh_tris;

  uint32_t ray_index = kInvalid;
  RayQ ray = {};
  f3 inv_d = {};
  Hit best = {};
  uint32_t alpha_seed = 0u, sp = 0u;
  int32_t cur = kDone;        // where the lane stands: an inner node (>= 0), a leaf it could not postpone (< 0), or nothing (kDone)
  uint32_t leaf_at = 0u, leaf_end = 0u;  // the postponed leaf: triangles [leaf_at, leaf_end) still to test
  bool busy = false;          // the lane holds a ray
  for (;;) {
    // ---- refill
    const unsigned long long idle_mask = __ballot(busy == false);
    const uint32_t idle_count = uint32_t(__popcll(idle_mask));
    if ((idle_count >= refill_lanes) || (idle_count == 64u)) {
      if (cursor < chunk_end) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(idle_mask >> 32u), __builtin_amdgcn_mbcnt_lo(uint32_t(idle_mask), 0u));
        const uint32_t take = min(idle_count, chunk_end - cursor);
        if ((busy == false) && (rank < take)) {
          ray_index = cursor + rank;
          const float4 a = ray_o_tmin[ray_index];
          const float4 b = ray_d_tmax[ray_index];
          alpha_seed = __float_as_uint(a.x) ^ (__float_as_uint(b.y) * 0x9e3779b9u) ^ (__float_as_uint(a.z) * 0x85ebca6bu) ^ (__float_as_uint(b.x) * 0xc2b2ae35u);
          ray = {{a.x, a.y, a.z}, a.w, {b.x, b.y, b.z}, b.w};
          inv_d = {__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z)};
          best = {0.0f, 0.0f, ray.tmax, kInvalid};
          sp = 0u;
          cur = scene.bvh_root;
          leaf_at = leaf_end = 0u;
          busy = true;
        }
        cursor += take;
      } else if (idle_count == 64u) {
        break;
      }
    }
    // ---- node phase
    for (;;) {
      const bool inner = (cur >= 0) && (cur != kDone);
      const uint32_t inner_count = uint32_t(__popcll(__ballot(inner)));
      if (inner_count == 0u)
        break;
      if ((inner_count < node_phase_lanes) && (__ballot(leaf_at < leaf_end) != 0ull))
        break;  // few lanes still walk while others hold leaves: test those first (the walkers resume in the next node phase)
      if (inner) {
        float4 lox, loy, loz, hix, hiy, hiz, cc;
        if (uint32_t(cur) < nodes.lds_count) {
          const float4* n = nodes.lds + uint32_t(cur) * 8u;
          lox = n[0], loy = n[1], loz = n[2], hix = n[3], hiy = n[4], hiz = n[5], cc = n[6];
        } else {
          const float4* n = nodes.global + uint32_t(cur) * 8u;
          lox = n[0], loy = n[1], loz = n[2], hix = n[3], hiy = n[4], hiz = n[5], cc = n[6];
        }
        float t0 = slab(f3{lox.x, loy.x, loz.x}, f3{hix.x, hiy.x, hiz.x}, ray.o, inv_d, ray.tmin, best.t);
        float t1 = slab(f3{lox.y, loy.y, loz.y}, f3{hix.y, hiy.y, hiz.y}, ray.o, inv_d, ray.tmin, best.t);
        float t2 = slab(f3{lox.z, loy.z, loz.z}, f3{hix.z, hiy.z, hiz.z}, ray.o, inv_d, ray.tmin, best.t);
        float t3 = slab(f3{lox.w, loy.w, loz.w}, f3{hix.w, hiy.w, hiz.w}, ray.o, inv_d, ray.tmin, best.t);
        int32_t c0 = __float_as_int(cc.x), c1 = __float_as_int(cc.y), c2 = __float_as_int(cc.z), c3 = __float_as_int(cc.w);
        t0 = (c0 == kBvhEmptyChild) ? kMaxFloat : t0;
        t1 = (c1 == kBvhEmptyChild) ? kMaxFloat : t1;
        t2 = (c2 == kBvhEmptyChild) ? kMaxFloat : t2;
        t3 = (c3 == kBvhEmptyChild) ? kMaxFloat : t3;
        sort_pair(t0, c0, t1, c1);
        sort_pair(t2, c2, t3, c3);
        sort_pair(t0, c0, t2, c2);
        sort_pair(t1, c1, t3, c3);
        sort_pair(t1, c1, t2, c2);
        if (t0 == kMaxFloat) {
          cur = sp ? stack.pop(sp) : kDone;
        } else {
          if (t3 < kMaxFloat)
            stack.push(sp, c3);
          if (t2 < kMaxFloat)
            stack.push(sp, c2);
          if (t1 < kMaxFloat)
            stack.push(sp, c1);
          cur = c0;
        }
        // a leaf: postpone it (if the lane holds none) and keep walking
        if ((cur < 0) && (leaf_at >= leaf_end)) {
          const uint32_t leaf = uint32_t(~cur);
          leaf_at = leaf >> 3, leaf_end = leaf_at + (leaf & 7u) + 1u;
          cur = sp ? stack.pop(sp) : kDone;
        }
      }
    }
    // ---- leaf phase: one triangle per trip for every lane that holds a leaf
    while (__ballot(leaf_at < leaf_end) != 0ull) {
      if (leaf_at < leaf_end) {
        const uint32_t i = leaf_at++;
        const float4 v0 = tris[i].v0_index;
        const float4 e1 = tris[i].e1_flags;
        const float4 e2 = tris[i].e2_mat;
        float u, v, t;
        if (triangle_test(v0, e1, e2, ray, best.t, u, v, t)) {
          const uint32_t flags = __float_as_uint(e1.w);
          const uint32_t tri_index = __float_as_uint(v0.w);
          if (((flags & kTriVoid) == 0u) && (((flags & kTriAlphaTested) == 0u) || (alpha_test_skips(scene, tri_index, __float_as_uint(e2.w), u, v, alpha_seed) == false)))
            best = {u, v, t, tri_index};
        }
      }
    }
    // a lane that stands on a second leaf (it held one when it got there) takes it now and pops on
    if (busy && (cur < 0)) {
      const uint32_t leaf = uint32_t(~cur);
      leaf_at = leaf >> 3, leaf_end = leaf_at + (leaf & 7u) + 1u;
      cur = sp ? stack.pop(sp) : kDone;
    }
    if (busy && (cur == kDone) && (leaf_at >= leaf_end)) {
      hits[ray_index] = make_float4(best.u, best.v, best.t, __uint_as_float(best.tri));
      busy = false;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Flat sweep, two rays per lane on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32). The one-ray sweep is VALU bound (about 30
// VALU per primitive and ray); here a lane carries rays i and i + 64 of a 128-ray chunk in the two halves of 64-bit
// register pairs, the primitive rows stay wave-uniform in SGPRs and are broadcast to both halves, so the affine part of
// the test (plane distance, hit point, the two parallelogram coordinates: 18 multiply-adds) costs 9 instructions per
// ray. The inside test is folded differently from flat_prim_test: a parallelogram needs |a - 1/2| <= 1/2 and
// |b - 1/2| <= 1/2 (one v_max3 with |.| modifiers, the -1/2 is a scalar subtraction on the row's constant), a triangle
// min3(a, b, 1 - a - b) >= 0. Only (t, primitive) are tracked in the loop; the coordinates of the winner are recomputed
// once after the sweep. Alpha-tested primitives (per-candidate random draws, scene_bsdf.hxx:128-144) take the one-ray
// code for each half so their draw order is the one of bvh_flat_closest.
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr uint32_t kFlat2Blocks = 2048u;  // 256 CUs x 8 blocks: more waves per SIMD overlap the queue traffic of one wave with the sweep of the others (measured 1024: 35.7, 2048: 39.3 Grays/s)

ETX_DEV v2f splat2(float v) {
  return v2f{v, v};
}

template <bool kFromCounter>
__global__ __launch_bounds__(kBlockSize) void k_trace_closest_flat2(const DScene scene_arg, const float4* __restrict__ ray_o_tmin, const float4* __restrict__ ray_d_tmax,
  float4* __restrict__ hits, uint32_t* __restrict__ counters, uint32_t active_counter, uint32_t fixed_count, unsigned long long* round_mirror, uint32_t round_tag, uint32_t pass_stat) {
  const DScene& scene = scene_arg;
  const uint32_t count = kFromCounter ? counters[active_counter] : fixed_count;
  if (kFromCounter && (blockIdx.x == 0) && (threadIdx.x == 0)) {
    round_housekeeping(counters, active_counter, count, pass_stat, round_mirror, round_tag);
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6u;
  const uint32_t wave_count = (gridDim.x * blockDim.x) >> 6u;
  const uint32_t prim_count = scene.flat_prim_count;
  ConstantFloats table = (ConstantFloats)(const void*)(scene.flat_prims);
  // software pipeline: the rays of the wave's next chunk are requested before the current chunk is swept, so the
  // queue traffic overlaps the VALU work (without it every wave alternates load / sweep / store phases and, with all
  // waves of the persistent grid in step, the memory pipe idles while the VALU works and vice versa)
  const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  uint32_t base = wave * 128u;
  float4 na0 = zero, nb0 = zero, na1 = zero, nb1 = zero;
  if (base + lane < count)
    na0 = ray_o_tmin[base + lane], nb0 = ray_d_tmax[base + lane];
  if (base + 64u + lane < count)
    na1 = ray_o_tmin[base + 64u + lane], nb1 = ray_d_tmax[base + 64u + lane];
  for (; base < count; base += wave_count * 128u) {
    const uint32_t i0 = base + lane, i1 = base + 64u + lane;
    const bool live0 = i0 < count, live1 = i1 < count;
    const float4 a0 = na0, b0 = nb0, a1 = na1, b1 = nb1;
    {
      const uint32_t n0 = i0 + wave_count * 128u, n1 = i1 + wave_count * 128u;
      if (n0 < count)
        na0 = ray_o_tmin[n0], nb0 = ray_d_tmax[n0];
      if (n1 < count)
        na1 = ray_o_tmin[n1], nb1 = ray_d_tmax[n1];
    }
    const v2f ox = {a0.x, a1.x}, oy = {a0.y, a1.y}, oz = {a0.z, a1.z};
    const v2f dx = {b0.x, b1.x}, dy = {b0.y, b1.y}, dz = {b0.z, b1.z};
    const float tmin0 = a0.w, tmin1 = a1.w;
    // a dead half gets an empty interval
    float best_t0 = live0 ? b0.w : -1.0f, best_t1 = live1 ? b1.w : -1.0f;
    uint32_t best_p0 = kInvalid, best_p1 = kInvalid;
    uint32_t seed0 = __float_as_uint(a0.x) ^ (__float_as_uint(b0.y) * 0x9e3779b9u) ^ i0;
    uint32_t seed1 = __float_as_uint(a1.x) ^ (__float_as_uint(b1.y) * 0x9e3779b9u) ^ i1;
#pragma unroll 2
    for (uint32_t k = 0; k < prim_count; ++k) {
      const FlatRow prim = load_flat_prim(table, k);
      const uint32_t flags = prim.flags;
      if (flags & kTriVoid)  // scalar branch
        continue;
      if (flags & kTriAlphaTested) {  // scalar branch; rare
        float a, b, t;
        const RayQ r0 = {{a0.x, a0.y, a0.z}, tmin0, {b0.x, b0.y, b0.z}, best_t0};
        if (flat_prim_test(prim, r0, best_t0, a, b, t) && (alpha_test_skips(scene, scene.flat_info[k].tri_a, prim.material, a, b, seed0) == false))
          best_t0 = t, best_p0 = k;
        const RayQ r1 = {{a1.x, a1.y, a1.z}, tmin1, {b1.x, b1.y, b1.z}, best_t1};
        if (flat_prim_test(prim, r1, best_t1, a, b, t) && (alpha_test_skips(scene, scene.flat_info[k].tri_a, prim.material, a, b, seed1) == false))
          best_t1 = t, best_p1 = k;
        continue;
      }
      const v2f den = splat2(prim.plane.x) * dx + splat2(prim.plane.y) * dy + splat2(prim.plane.z) * dz;
      const v2f num = splat2(prim.plane.x) * ox + splat2(prim.plane.y) * oy + splat2(prim.plane.z) * oz + splat2(prim.plane.w);
      const v2f inv = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
      const v2f t = -num * inv;
      const v2f px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;
      bool in0, in1;
      if (flags & kTriQuad) {  // scalar branch
        const float wa = prim.row_a.w - 0.5f, wb = prim.row_b.w - 0.5f;  // SALU
        const v2f a = splat2(prim.row_a.x) * px + splat2(prim.row_a.y) * py + splat2(prim.row_a.z) * pz + splat2(wa);
        const v2f b = splat2(prim.row_b.x) * px + splat2(prim.row_b.y) * py + splat2(prim.row_b.z) * pz + splat2(wb);
        in0 = fmaxf(fabsf(a.x), fabsf(b.x)) <= 0.5f;
        in1 = fmaxf(fabsf(a.y), fabsf(b.y)) <= 0.5f;
      } else {
        const v2f a = splat2(prim.row_a.x) * px + splat2(prim.row_a.y) * py + splat2(prim.row_a.z) * pz + splat2(prim.row_a.w);
        const v2f b = splat2(prim.row_b.x) * px + splat2(prim.row_b.y) * py + splat2(prim.row_b.z) * pz + splat2(prim.row_b.w);
        const v2f e = (splat2(1.0f) - a) - b;
        in0 = fminf(fminf(a.x, b.x), e.x) >= 0.0f;
        in1 = fminf(fminf(a.y, b.y), e.y) >= 0.0f;
      }
      // a ray parallel to the plane gives t = inf / nan: both fail the explicit comparisons on t
      if ((t.x >= tmin0) && (t.x <= best_t0) && in0)
        best_t0 = t.x, best_p0 = k;
      if ((t.y >= tmin1) && (t.y <= best_t1) && in1)
        best_t1 = t.y, best_p1 = k;
    }
    // coordinates of the winners (per-lane primitive index: ordinary loads, once per ray)
    auto resolve = [&](uint32_t prim_index, const float4& ro, const float4& rd, float t, float t_max) -> float4 {
      if (prim_index == kInvalid)
        return make_float4(0.0f, 0.0f, t_max, __uint_as_float(kInvalid));
      const FlatPrim& pr = scene.flat_prims[prim_index];
      const float4 ra = pr.row_a, rb = pr.row_b;
      const f3 x = f3{ro.x, ro.y, ro.z} + f3{rd.x, rd.y, rd.z} * t;
      const float a = ra.x * x.x + ra.y * x.y + ra.z * x.z + ra.w;
      const float b = rb.x * x.x + rb.y * x.y + rb.z * x.z + rb.w;
      const Hit h = flat_resolve(scene, prim_index, a, b, t);
      return make_float4(h.u, h.v, h.t, __uint_as_float(h.tri));
    };
    if (live0)
      hits[i0] = resolve(best_p0, a0, b0, best_t0, b0.w);
    if (live1)
      hits[i1] = resolve(best_p1, a1, b1, best_t1, b1.w);
  }
}

static uint32_t flat2_blocks(uint32_t items) {  // 512 rays per 256-thread block and loop iteration
  const uint32_t limit = etxh::tuning_knob("ETX_HIP_DEBUG_BLOCKS", kFlat2Blocks);
  return max(1u, min(limit, (items + 2u * kBlockSize - 1u) / (2u * kBlockSize)));
}

// Persistent grid of the BVH kernel: as many workgroups as the device keeps resident (4 per CU at 40 KB of LDS each), fewer
// for small queues so that every wavefront still owns several 64-ray rows to refill from.
static uint32_t bvh_blocks(uint32_t items) {
  const uint32_t limit = etxh::tuning_knob("ETX_HIP_DEBUG_BLOCKS", 256u * 4u);
  return max(1u, min(limit, (items + 4u * kBlockSize - 1u) / (4u * kBlockSize)));
}

static uint32_t lds_limit() {  // experiments: ETX_HIP_LDS_NODES caps the staged part of the tree (0 = everything through L2)
  static const uint32_t limit = etxh::tuning_knob("ETX_HIP_LDS_NODES", kLdsNodes);
  return limit;
}

// Variant by the stack the scene's tree needs (three entries per BVH4 level): 16 entries -> 24 KB of LDS per workgroup and six resident
// workgroups per CU, 32 entries otherwise; a deep tree (> ~40 000 triangles: bound above 32) the checked stack, 16 entries in LDS, the rest in the
// global spill rows (dev_bvh.h ShortLaneStack; a million triangles: 13.1 vs 12.8 Msamples/s with 32 entries in LDS).
// ---------------------------------------------------------------------------------------------------------------
// Flat sweep with the affine part on the MATRIX cores (north_star: "MFMA ... for the per-vertex 3x3 frame transforms if rocprof shows it paying
// off"). The sweep's test of primitive p against ray r is three affine forms of the hit point (plane distance and the two
// parallelogram coordinates: FlatPrim rows `plane`, `row_a`, `row_b`, each [x y z w]) - and since the hit point is o + t d, each is
// row.[o;1] + t (row.[d;0]): SIX dot products per (primitive, ray) that do not depend on one another. That is a small dense product: rows of four
// primitives (16 rows, the fourth of each primitive zero) x the [o;1] and [d;0] columns of the 64 rays of a wavefront = 7 x
// v_mfma_f32_16x16x1_4b_f32 (K = 1: one ray component per instruction, the zero w of d skipped), 224 matrix-pipe cycles for 256 (primitive, ray) tests.
// What is left on the VALU per test: t = -num / den (v_rcp + v_mul), a = Ao + t Ad, b = Bo + t Bd (2 v_fma), the inside minimum (5), three compares and
// four selects: 16 instructions where flat_prim_test + the best-hit update issue 29.
// Layout (CDNA3/4 ISA, V_MFMA_F32_16X16X1_4B_F32: four independent 16x16 blocks; lane l feeds A[l % 16] and B[l % 16] of block l / 16 and receives,
// in accumulator 4 b + r, D_b[4 (l / 16) + r][l % 16]): block b = rays 16 b .. 16 b + 15 of the wavefront, so a lane supplies ITS OWN ray as B and row
// l % 16 of the current four primitives as A (read from LDS), and gets back the six forms of ONE primitive (slot g = l / 16 of the four) for FOUR rays
// (l % 16 of every block). Every lane therefore tests one primitive against four rays per step and keeps four running best hits; after the last
// step the four lanes that share a ray (l, l ^ 16, l ^ 32, l ^ 48) fold their bests (ties go to the higher primitive index, as the sequential `t <=
// best` of bvh_flat_closest gives them) and lane l picks block l / 16: its own ray.
// Scenes with alpha-tested primitives (per-candidate random draws in sweep order) keep the VALU sweep (DScene::bvh_flat bit 1).
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr uint32_t kMfmaSets = kFlatSweepMaxTriangles / 4u;  // steps of four primitives

struct MfmaTables {  // LDS
  float4 rows[kMfmaSets * 16u];   // [step][row]: plane / row_a / row_b / 0 of primitive 4 step + row / 4 (all zero: void, degenerate or padding - never hit)
  float quad[kMfmaSets * 4u];     // per primitive: 1 = parallelogram, 0 = triangle
  uint32_t flags[kMfmaSets * 4u];
};

ETX_DEV void mfma_stage_tables(const DScene& scene, MfmaTables& tables) {
  const uint32_t prim_count = scene.flat_prim_count;
  const float* source = reinterpret_cast<const float*>(scene.flat_prims);
  for (uint32_t i = threadIdx.x; i < kMfmaSets * 16u; i += blockDim.x) {
    const uint32_t prim = (i >> 4u) * 4u + ((i & 15u) >> 2u), which = i & 3u;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if ((prim < prim_count) && (which < 3u)) {
      const float* t = source + size_t(prim) * 16u;
      if ((__float_as_uint(t[12]) & kTriVoid) == 0u)
        v = make_float4(t[which * 4u + 0u], t[which * 4u + 1u], t[which * 4u + 2u], t[which * 4u + 3u]);
    }
    tables.rows[i] = v;
  }
  for (uint32_t i = threadIdx.x; i < kMfmaSets * 4u; i += blockDim.x) {
    const uint32_t flags = (i < prim_count) ? __float_as_uint(source[size_t(i) * 16u + 12u]) : 0u;
    tables.flags[i] = flags;
    tables.quad[i] = (flags & kTriQuad) ? 1.0f : 0.0f;
  }
  __syncthreads();
}

// Closest primitive of the lane's own ray (origin / tmin in `a`, direction / tmax in `b`; a lane without a ray passes tmax < tmin). Must be called by
// ALL 64 lanes of the wavefront (the matrix instructions take their operands from every lane).
ETX_DEV void mfma_sweep(const MfmaTables& tables, uint32_t steps, const float4& a, const float4& b, float& out_t, float& out_a, float& out_b, int32_t& out_prim) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t col = lane & 15u, slot = lane >> 4u;
  float tmin4[4], best_t[4], best_a[4], best_b[4];
  int32_t best_p[4];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    tmin4[k] = __shfl(a.w, int(16u * k + col));
    best_t[k] = __shfl(b.w, int(16u * k + col));
    best_a[k] = best_b[k] = 0.0f;
    best_p[k] = -1;
  }
  // software pipeline: the seven matrix instructions of step s + 1 are issued before the VALU work on the results of step s, so the (separate)
  // matrix pipe computes while this wavefront's own VALU instructions issue - not only those of the other wavefronts of the SIMD
  const floatx16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  auto forms = [&](uint32_t step, floatx16& with_o, floatx16& with_d) {
    const float4 row = tables.rows[step * 16u + col];
    with_o = __builtin_amdgcn_mfma_f32_16x16x1f32(row.x, a.x, zero16, 0, 0, 0);
    with_d = __builtin_amdgcn_mfma_f32_16x16x1f32(row.x, b.x, zero16, 0, 0, 0);
    with_o = __builtin_amdgcn_mfma_f32_16x16x1f32(row.y, a.y, with_o, 0, 0, 0);
    with_d = __builtin_amdgcn_mfma_f32_16x16x1f32(row.y, b.y, with_d, 0, 0, 0);
    with_o = __builtin_amdgcn_mfma_f32_16x16x1f32(row.z, a.z, with_o, 0, 0, 0);
    with_d = __builtin_amdgcn_mfma_f32_16x16x1f32(row.z, b.z, with_d, 0, 0, 0);
    with_o = __builtin_amdgcn_mfma_f32_16x16x1f32(row.w, 1.0f, with_o, 0, 0, 0);
  };
  auto tests = [&](uint32_t step, const floatx16& with_o, const floatx16& with_d) {
    const float quad = tables.quad[step * 4u + slot];
    const float tri = 1.0f - quad;
    const int32_t prim = int32_t(step * 4u + slot);
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      const float num = with_o[4u * k + 0u], a_o = with_o[4u * k + 1u], b_o = with_o[4u * k + 2u];
      const float den = with_d[4u * k + 0u], a_d = with_d[4u * k + 1u], b_d = with_d[4u * k + 2u];
      const float t = -num * __builtin_amdgcn_rcpf(den);  // den = 0 (parallel, void, padding): inf / nan fail the comparisons below
      const float ca = fmaf(t, a_d, a_o), cb = fmaf(t, b_d, b_o);
      const float e = fmaf(-tri, ca, 1.0f - cb);
      const float f = fmaf(-quad, ca, 1.0f);
      const float inside = fminf(fminf(fminf(ca, cb), e), f);
      const bool hit = bool(int(t >= tmin4[k]) & int(t <= best_t[k]) & int(inside >= 0.0f));  // no short circuit: four selects, no branch
      best_t[k] = hit ? t : best_t[k];
      best_a[k] = hit ? ca : best_a[k];
      best_b[k] = hit ? cb : best_b[k];
      best_p[k] = hit ? prim : best_p[k];
    }
  };
  // two result sets, used alternately (a loop that handed `next` over to `current` copied 32 registers per step)
  floatx16 o0, d0, o1 = zero16, d1 = zero16;
  forms(0u, o0, d0);
  for (uint32_t step = 0; step < steps; step += 2u) {  // trip counts and branches are wave-uniform
    const bool second = step + 1u < steps;
    if (second)
      forms(step + 1u, o1, d1);
    tests(step, o0, d0);
    if (second) {
      if (step + 2u < steps)
        forms(step + 2u, o0, d0);
      tests(step + 1u, o1, d1);
    }
  }
  // the four lanes of a ray: nearer wins, equal distances go to the higher primitive index (-1 = none is the lowest)
#pragma unroll
  for (uint32_t distance = 16u; distance <= 32u; distance <<= 1u) {
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      const float ot = __shfl_xor(best_t[k], int(distance)), oa = __shfl_xor(best_a[k], int(distance)), ob = __shfl_xor(best_b[k], int(distance));
      const int32_t op = __shfl_xor(best_p[k], int(distance));
      const bool take = bool(int(op >= 0) & (int(best_p[k] < 0) | int(ot < best_t[k]) | (int(ot == best_t[k]) & int(op > best_p[k]))));
      best_t[k] = take ? ot : best_t[k], best_a[k] = take ? oa : best_a[k], best_b[k] = take ? ob : best_b[k], best_p[k] = take ? op : best_p[k];
    }
  }
  out_t = (slot == 0u) ? best_t[0] : ((slot == 1u) ? best_t[1] : ((slot == 2u) ? best_t[2] : best_t[3]));
  out_a = (slot == 0u) ? best_a[0] : ((slot == 1u) ? best_a[1] : ((slot == 2u) ? best_a[2] : best_a[3]));
  out_b = (slot == 0u) ? best_b[0] : ((slot == 1u) ? best_b[1] : ((slot == 2u) ? best_b[2] : best_b[3]));
  out_prim = (slot == 0u) ? best_p[0] : ((slot == 1u) ? best_p[1] : ((slot == 2u) ? best_p[2] : best_p[3]));
}

// (two workgroups per CU as the minimum occupancy = at most 256 registers per lane: with that bound the compiler keeps the matrix results in
// ordinary VGPRs - "VGPR form" - instead of accumulation registers it would have to copy out one v_accvgpr_read at a time, 8 extra VALU per test)
template <bool kFromCounter, bool kCross>
__global__ __launch_bounds__(kBlockSize, 2) void k_trace_closest_mfma(const DScene scene_arg, const float4* ray_o_tmin, const float4* ray_d_tmax, float4* __restrict__ hits,
  uint32_t* __restrict__ counters, uint32_t active_counter, uint32_t fixed_count, unsigned long long* round_mirror, uint32_t round_tag, uint32_t pass_stat, PathSet set = {},
  uint32_t cross_mode = kCrossNone, unsigned long long* block_stats = nullptr) {
  __shared__ MfmaTables s_tables;
  const DScene& scene = scene_arg;
  const uint32_t count = kFromCounter ? counters[active_counter] : fixed_count;
  if (kFromCounter && (blockIdx.x == 0) && (threadIdx.x == 0)) {
    round_housekeeping(counters, active_counter, count, pass_stat, round_mirror, round_tag);
  }
  mfma_stage_tables(scene, s_tables);
  const uint32_t steps = (scene.flat_prim_count + 3u) / 4u;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long crossed_queries = 0ull;
  for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x - lane; base < count; base += stride) {  // wave-uniform trip count: every lane runs every sweep
    const uint32_t i = base + lane;
    const bool live = i < count;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = make_float4(0.0f, 0.0f, 1.0f, -1.0f);  // no ray: an empty interval
    if (live)
      a = ray_o_tmin[i], b = ray_d_tmax[i];
    float t, ca, cb;
    int32_t prim;
    mfma_sweep(s_tables, steps, a, b, t, ca, cb, prim);
    if (kCross) {
      uint32_t crossings = 0u, medium = kInvalid;
      float crossed = 0.0f;
      f3 origin = {a.x, a.y, a.z};
      const f3 direction = {b.x, b.y, b.z};
      bool crossing = live && (prim >= 0) && ((s_tables.flags[prim >= 0 ? prim : 0] & kTriBoundary) != 0u);
      if (crossing) {
        medium = set.meta[i].z;
        crossing = medium == kInvalid;
      }
      while (__any(crossing)) {  // wave-uniform: the matrix instructions need every lane; a lane that is not crossing sweeps an empty interval
        float4 na = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nb = make_float4(0.0f, 0.0f, 1.0f, -1.0f);
        if (crossing) {
          const Hit h = flat_resolve(scene, uint32_t(prim), ca, cb, t);
          const etx_abi_triangle& tri = scene.triangles[h.tri];
          const etx_abi_material& mat = scene.materials[tri.material_index];
          medium = (dot(ld3(tri.geo_n), direction) < 0.0f) ? mat.int_medium : mat.ext_medium;
          crossed += t;
          origin = shading_pos(scene, tri, barycentrics(h.u, h.v), direction);
          crossings += 1u;
          na = mk4(origin, kRayEpsilon), nb = mk4(direction, kMaxFloat);
        }
        float nt, nca, ncb;
        int32_t nprim;
        mfma_sweep(s_tables, steps, na, nb, nt, nca, ncb, nprim);
        if (crossing) {
          t = nt, ca = nca, cb = ncb, prim = nprim;
          // the next boundary is crossed too only by a path that is (still) in no medium
          crossing = (prim >= 0) && ((s_tables.flags[prim] & kTriBoundary) != 0u) && (medium == kInvalid) && (crossings < 8u);
        }
      }
      crossed_queries += crossings;
      if (crossings != 0u) {
        const_cast<float4*>(ray_o_tmin)[i] = mk4(origin, kRayEpsilon);
        const_cast<float4*>(ray_d_tmax)[i] = mk4(direction, kMaxFloat);
        uint4 meta = set.meta[i];
        meta.z = medium;
        if (cross_mode == kCrossBdpt) {
          Sampler stream;
          stream.seed = meta.x;
          for (uint32_t k = 0; k < crossings * 6u; ++k)
            (void)stream.next();
          meta.x = stream.seed;
        }
        set.meta[i] = meta;
        if (cross_mode == kCrossVcm) {
          float4 mis = set.mis[i];
          mis.w += crossed;
          set.mis[i] = mis;
        }
      }
    }
    if (live) {
      if (prim < 0) {
        hits[i] = make_float4(0.0f, 0.0f, t, __uint_as_float(kInvalid));  // t = the tmax of the (last) segment swept: mfma_sweep starts from it
      } else {
        const Hit h = flat_resolve(scene, uint32_t(prim), ca, cb, t);
        hits[i] = make_float4(h.u, h.v, h.t, __uint_as_float(h.tri));
      }
    }
  }
  if (kCross) {
    __shared__ unsigned long long s_stat;
    Pipeline stats_only = {};
    stats_only.block_stats = block_stats;
    block_stat_add(stats_only, kBlockStatCrossings, crossed_queries, &s_stat);
  }
}

template <bool kFromCounter>
static void launch_bvh_kernel(hipStream_t stream, const DScene& scene, const float4* ray_o_tmin, const float4* ray_d_tmax, float4* hits, uint32_t* counters, uint32_t active_counter,
  uint32_t items, unsigned long long* round_mirror, uint32_t round_tag, uint32_t pass_stat) {
  static const uint32_t refill = etxh::tuning_knob("ETX_HIP_REFILL_LANES", kRefillLanes);
  static const uint32_t node_phase = etxh::tuning_knob("ETX_HIP_NODE_PHASE_LANES", kNodePhaseLanes);
  const dim3 grid(bvh_blocks(items)), block(kBlockSize);
  const uint32_t fixed_count = kFromCounter ? 0u : items;
#define ETX_LAUNCH_BVH(STACK, DEEP)                                                                                                                                                      \
  hipLaunchKernelGGL((k_trace_closest_bvh<kFromCounter, STACK, 64u, DEEP>), grid, block, 0, stream, scene, ray_o_tmin, ray_d_tmax, hits, counters, active_counter, fixed_count, round_mirror, \
    round_tag, lds_limit(), refill, node_phase, pass_stat)
  const uint32_t need = scene.bvh_stack_need;
  if (need > kStackDepth)
    ETX_LAUNCH_BVH(kShortStackDepth, true);
  else if (need <= 16u)
    ETX_LAUNCH_BVH(16u, false);
  else if (need <= 24u)
    ETX_LAUNCH_BVH(24u, false);
  else
    ETX_LAUNCH_BVH(kStackDepth, false);
#undef ETX_LAUNCH_BVH
}

void launch_trace_closest(hipStream_t stream, const Pipeline& p, uint32_t set, uint32_t active_counter, uint32_t max_items, bool flat, unsigned long long* round_mirror, uint32_t round_tag, uint32_t pass_stat,
  uint32_t cross_mode) {
  uint32_t blocks = max(1u, min(kPersistentBlocks, (min(p.capacity, max_items) + kBlockSize - 1) / kBlockSize));
  // opt-in (ETX_HIP_DEBUG_FLAGS bit 64): alone on the device the two-ray sweep is 10 % faster (39.3 vs 35.7 Grays/s on 2 M incoherent
  // rays), inside the 4-lane pipeline its fatter waves lose against the co-running kernels (51.5 vs 46.6 us per launch) - DESIGN.md 3
  // opt-in (debug flag 128): the sweep with its affine part on the matrix cores (k_trace_closest_mfma; measured in profiles/round6_ab_mfma_sweep.txt)
  const bool mfma = flat && (p.debug_flags & 128u) && ((p.scene.bvh_flat & 2u) == 0u);
  if (mfma && (cross_mode != kCrossNone) && (p.scene.boundary_materials != 0u))
    hipLaunchKernelGGL((k_trace_closest_mfma<true, true>), dim3(blocks), dim3(kBlockSize), 0, stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits, p.counters, active_counter, 0u, round_mirror,
      round_tag, pass_stat, p.paths[set], cross_mode, p.block_stats);
  else if (mfma)
    hipLaunchKernelGGL((k_trace_closest_mfma<true, false>), dim3(blocks), dim3(kBlockSize), 0, stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits, p.counters, active_counter, 0u, round_mirror,
      round_tag, pass_stat);
  else if (flat && (p.debug_flags & 64u))
    hipLaunchKernelGGL((k_trace_closest_flat2<true>), dim3(flat2_blocks(min(p.capacity, max_items))), dim3(kBlockSize), 0, stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits,
      p.counters, active_counter, 0u, round_mirror, round_tag, pass_stat);
  else if (flat && (cross_mode != kCrossNone) && (p.scene.boundary_materials != 0u))
    hipLaunchKernelGGL((k_trace_closest<true, true, true>), dim3(blocks), dim3(kBlockSize), 0, stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits, p.counters, active_counter, 0u, round_mirror, round_tag,
      lds_limit(), pass_stat, p.paths[set], cross_mode, p.block_stats);
  else if (flat)
    hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(blocks), dim3(kBlockSize), 0, stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits, p.counters, active_counter, 0u, round_mirror, round_tag, lds_limit(), pass_stat);
  else
    launch_bvh_kernel<true>(stream, p.scene, p.paths[set].ray_o_tmin, p.paths[set].ray_d_tmax, p.hits, p.counters, active_counter, min(p.capacity, max_items), round_mirror, round_tag, pass_stat);
}

// ---------------------------------------------------------------------------------------------------------------
// Shadow kernel: segment queue -> transmittance -> film atomics (Raytracing::trace_transmittance, rt.cxx:468-579, plus
// the accumulation the callers do: vcm_cpu.cxx:148-153 light splats, vcm_shared.hxx:1049-1053 camera gathers).
// Algorithmic traffic: 48 B request in, 12 B of float atomics out for visible segments.
// LDS: the per-lane stacks (32 KB) + the top kShadowLdsNodes nodes of the tree (8 KB): 40 KB per workgroup, three to four resident
// workgroups per CU. (256 staged nodes = 64 KB left two: the kernel waits on the node fetches below the staged levels, and more
// wavefronts hide more of that than more staged levels save.)
constexpr uint32_t kShadowLdsNodes = 64u;
// The opaque kernel (no boundaries, no density grids: 94 VGPRs instead of 137) stages no nodes: 32 KB of stacks = five workgroups per CU,
// which its registers allow as well. configs[3] (47 M segments on a 102 k-triangle tree per iteration), six lanes: general kernel 26.5,
// opaque with 64 staged nodes 27.6, without 28.7 Msamples/s (interleaved A/B in one session, tools/gpu_calls/gpu_r3n.sh).
#if !defined(ETX_SHADOW_OPAQUE_NODES)
#define ETX_SHADOW_OPAQUE_NODES 0u
#endif
// kOpaque (tree scenes only): no Class::Boundary material and no density grid in the scene - bvh_transmittance_opaque
#if !defined(ETX_SHADOW_OPAQUE_STACK)
#define ETX_SHADOW_OPAQUE_STACK kShortStackDepth
#endif
// waves per SIMD the opaque kernel is compiled for: 7 = 72 VGPRs (three spilled), LDS for sixteen stack entries per lane. configs[3], same
// session: 98 VGPRs / five waves 28.7, 80 / six 29.0, 72 / seven 29.2 Msamples/s (tools/gpu_calls/gpu_r3o.sh)
#if !defined(ETX_SHADOW_OPAQUE_WAVES)
#define ETX_SHADOW_OPAQUE_WAVES 7
#endif
template <bool kFlat, bool kDeep = false, bool kOpaque = false>
__global__ __launch_bounds__(kBlockSize, kOpaque ? ETX_SHADOW_OPAQUE_WAVES : 1) void k_trace_shadow(Pipeline p) {
  constexpr bool kShort = kOpaque && (uint32_t(ETX_SHADOW_OPAQUE_STACK) == kShortStackDepth);  // the opaque kernel: LDS for 16 entries per lane, the rest spills
  __shared__ int32_t s_stack[kFlat ? 1 : (kShort ? kShortStackDepth : kStackDepth) * kBlockSize];
  constexpr uint32_t kLdsNodes = kOpaque ? uint32_t(ETX_SHADOW_OPAQUE_NODES) : kShadowLdsNodes;
  __shared__ float4 s_nodes[(kFlat || (kLdsNodes == 0u)) ? 1 : kLdsNodes * 8u];
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntShadow], p.shadow.capacity);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t stride = gridDim.x * blockDim.x;
  typedef TraversalStack<kDeep || kShort, kShort ? kShortStackDepth : kStackDepth> StackKind;
  const typename StackKind::Type stack = StackKind::make(scene, s_stack + (kFlat ? 0u : threadIdx.x), kBlockSize);  // a flat scene only touches it when a segment crosses > 4 boundaries
  uint32_t splats = 0;
  BvhNodes nodes = global_nodes(scene);
  if ((kFlat == false) && (kLdsNodes != 0u) && (blockIdx.x * blockDim.x < count))
    nodes = stage_nodes(scene, s_nodes, kLdsNodes);
  for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x - lane; base < count; base += stride) {
    const uint32_t i = base + lane;
    f3 value = mk3(0.0f);
    uint32_t target = 0xffffffffu - lane;  // idle lanes: a target nobody shares
    if (i < count) {
      const float4 a = p.shadow.p0_medium[i];
      const float4 b = p.shadow.p1_target[i];
      uint32_t alpha_seed = __float_as_uint(a.x) ^ (__float_as_uint(b.y) * 0x9e3779b9u) ^ (__float_as_uint(a.z) * 0x85ebca6bu) ^ (__float_as_uint(b.x) * 0xc2b2ae35u);  // of the segment itself, not of its queue slot
      f3 tr = mk3(1.0f);
      if (kOpaque)
        tr = bvh_transmittance_opaque(scene, nodes, scene.bvh_tris, scene.bvh_root, stack, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, __float_as_uint(a.w), p.shadow.value[i].w, alpha_seed);
      else if ((p.debug_flags & 4u) == 0u)
        tr = bvh_transmittance(scene, nodes, scene.bvh_tris, scene.bvh_root, stack, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, __float_as_uint(a.w), p.shadow.value[i].w, alpha_seed);
      target = __float_as_uint(b.w);
      if ((tr.x > kEpsilon) || (tr.y > kEpsilon) || (tr.z > kEpsilon)) {  // SpectralResponse::is_zero
        const float4 v = p.shadow.value[i];
        value = tr * f3{v.x, v.y, v.z};
        if (film_value_ok(p, value) == false)
          value = mk3(0.0f);
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

#if defined(ETX_NO_OPAQUE_SHADOW)
  const bool opaque = false;
#else
  const bool opaque = (p.scene.boundary_materials == 0u) && (p.scene.heterogeneous_mediums == 0u) && ((p.debug_flags & 4u) == 0u);
#endif
  if (flat)
    hipLaunchKernelGGL(k_trace_shadow<true>, dim3(blocks), dim3(kBlockSize), 0, stream, p);
  else if (p.scene.bvh_stack_need > kStackDepth) {
    if (opaque)
      hipLaunchKernelGGL((k_trace_shadow<false, true, true>), dim3(blocks), dim3(kBlockSize), 0, stream, p);
    else
      hipLaunchKernelGGL((k_trace_shadow<false, true>), dim3(blocks), dim3(kBlockSize), 0, stream, p);
  } else if (opaque)
    hipLaunchKernelGGL((k_trace_shadow<false, false, true>), dim3(blocks), dim3(kBlockSize), 0, stream, p);
  else
    hipLaunchKernelGGL(k_trace_shadow<false>, dim3(blocks), dim3(kBlockSize), 0, stream, p);
}

void launch_trace_rays(hipStream_t stream, const DScene& scene, const float4* ray_o_tmin, const float4* ray_d_tmax, float4* hits, uint32_t count, bool flat, uint32_t debug_flags) {
  uint32_t blocks = min(kPersistentBlocks, (count + kBlockSize - 1) / kBlockSize);
  if (blocks == 0)
    return;
  DScene limited = scene;  // experiments: ETX_HIP_DEBUG_PRIMS caps the primitive count of the flat sweep (cost per primitive vs memory floor)
  const uint32_t prim_cap = etxh::tuning_knob("ETX_HIP_DEBUG_PRIMS", 0u);  // 0 = no cap
  if ((prim_cap != 0u) && (prim_cap < limited.flat_prim_count))
    limited.flat_prim_count = prim_cap;
  const bool two_ray_sweep = (debug_flags & 64u) != 0u;  // etx_hip_set_debug_flags: the packed two-ray sweep (kept, measured, not the default)
  if (flat && (debug_flags & 128u) && ((limited.bvh_flat & 2u) == 0u))
    hipLaunchKernelGGL((k_trace_closest_mfma<false, false>), dim3(blocks), dim3(kBlockSize), 0, stream, limited, ray_o_tmin, ray_d_tmax, hits, nullptr, 0u, count, nullptr, 0u, 0u);
  else if (flat && two_ray_sweep)
    hipLaunchKernelGGL((k_trace_closest_flat2<false>), dim3(flat2_blocks(count)), dim3(kBlockSize), 0, stream, limited, ray_o_tmin, ray_d_tmax, hits, nullptr, 0u, count, nullptr, 0u, 0u);
  else if (flat)
    hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(blocks), dim3(kBlockSize), 0, stream, limited, ray_o_tmin, ray_d_tmax, hits, nullptr, 0u, count, nullptr, 0u, lds_limit(), 0u);
  else
    launch_bvh_kernel<false>(stream, limited, ray_o_tmin, ray_d_tmax, hits, nullptr, 0u, count, nullptr, 0u, 0u);
}

}  // namespace etxd
