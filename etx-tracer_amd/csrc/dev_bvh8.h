// dev_bvh8.h - the eight-wide tree with 8-bit child boxes (dev_scene.h Bvh8Node): one 128-byte fetch decides eight children.
//
// Why: the traversal kernels of tree scenes wait on the chain of dependent node fetches; the same binned-SAH tree
// collapsed to eight children per node needs 38 % fewer node visits per subsurface walk segment and a third fewer for incoherent
// rays (tools/bvh_study.py, profiles/round3_bvh_format_study.txt) at the node size of today's four-wide packet.
//
// The child boxes are offsets q in [0, 255] on a per-node grid: box = origin + q * 2^e per axis, rounded outwards by the host
// (host_scene.cpp encode_bvh8) with a margin of a few units in the last place, so the folded slab test below - the ray is moved
// into the node's frame once, every bound then costs one conversion and one fused multiply-add - cannot cut into the exact box.
// The traversal visits the nearest hit child first and pushes the others as they come (no sorting network for eight; +0.2-6 % of
// visits against sorted pushes). bvh8_visit is host + device code: etx_hip_host_bvh8_stats walks the ENCODED tree on the host
// through the very function the kernels call (tests/test_host_bvh8.py: the hits of the four-wide tree, ray by ray).
//
// STATUS (round 3): opt-in (etx_hip_set_bvh_builder(... | ETX_HIP_BVH_WIDE)); the kernels that read it have compiled for gfx950 but
// have not run on a device yet - the default tree is the four-wide one of dev_bvh.h.
#pragma once

#include "dev_bvh.h"

namespace etxd {

ETX_HD float bvh8_bits_to_float(uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(bits);
#else
  float f;
  memcpy(&f, &bits, 4);
  return f;
#endif
}

ETX_HD float bvh8_byte(uint32_t word, uint32_t k) {  // v_cvt_f32_ubyte<k>
  return float((word >> (8u * k)) & 0xffu);
}

// Reciprocal direction for the folded slab test. A zero component would make step and base infinite and their sum NaN (the axis would
// drop out of the test: correct but hundreds of extra visits for an axis-parallel ray); a huge finite reciprocal keeps the axis in -
// both bounds land at -huge / +huge when the origin is between them and on one side otherwise.
ETX_HD float bvh8_reciprocal(float d) {
  const float kHuge = 1.0e30f;
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_rcpf(d);
#else
  const float r = 1.0f / d;
#endif
  return (fabsf(r) < kHuge) ? r : ((d < 0.0f) ? -kHuge : kHuge);
}

// One node in registers: eight 16-byte words (Bvh8Node's layout)
struct Bvh8Words {
  float4 frame;        // origin.xyz, exponents (u32 bits: ex | ey << 8 | ez << 16, biased float exponents of the grid steps)
  uint4 lo_xy;         // qlo_x[0..7], qlo_y[0..7]
  uint4 lo_z_hi_x;     // qlo_z[0..7], qhi_x[0..7]
  uint4 hi_yz;         // qhi_y[0..7], qhi_z[0..7]
  int4 child_a, child_b;
};

ETX_HD Bvh8Words bvh8_load(const uint4* node) {
  Bvh8Words w;
  const uint4 f = node[0];
  w.frame = make_float4(bvh8_bits_to_float(f.x), bvh8_bits_to_float(f.y), bvh8_bits_to_float(f.z), bvh8_bits_to_float(f.w));
  w.lo_xy = node[1], w.lo_z_hi_x = node[2], w.hi_yz = node[3];
  const uint4 a = node[4], b = node[5];
  w.child_a = make_int4(int32_t(a.x), int32_t(a.y), int32_t(a.z), int32_t(a.w));
  w.child_b = make_int4(int32_t(b.x), int32_t(b.y), int32_t(b.z), int32_t(b.w));
  return w;
}

// The ray in the frame of a node: t(q) = q * step + base per axis
struct Bvh8RayFrame {
  f3 step, base;
  float slack;  // what rounding can move a slab distance by: the folded form t = q * step + base loses the bits of `base`, which grows with the
                // distance of the ray origin from the node (a camera far outside the scene): a few ulps of |base| are given back to the interval.
                // The host's outward rounding (encode_bvh8) covers origins inside the scene's reach; this covers the rest.
};

ETX_HD Bvh8RayFrame bvh8_ray_frame(const float4& frame, const f3& ray_o, const f3& inv_d) {
  uint32_t exps;
#if defined(__HIP_DEVICE_COMPILE__)
  exps = __float_as_uint(frame.w);
#else
  memcpy(&exps, &frame.w, 4);
#endif
  const f3 scale = {bvh8_bits_to_float((exps & 0xffu) << 23u), bvh8_bits_to_float(((exps >> 8u) & 0xffu) << 23u), bvh8_bits_to_float(((exps >> 16u) & 0xffu) << 23u)};
  const f3 base = {(frame.x - ray_o.x) * inv_d.x, (frame.y - ray_o.y) * inv_d.y, (frame.z - ray_o.z) * inv_d.z};
  // an axis the ray is parallel to has a clamped reciprocal (bvh8_reciprocal): its bounds sit at +-1e30 x distance and decide by sign alone
  const float kClamped = 1.0e29f;
  const float magnitude = ((fabsf(inv_d.x) < kClamped) ? fabsf(base.x) : 0.0f) + ((fabsf(inv_d.y) < kClamped) ? fabsf(base.y) : 0.0f) + ((fabsf(inv_d.z) < kClamped) ? fabsf(base.z) : 0.0f);
  return {{scale.x * inv_d.x, scale.y * inv_d.y, scale.z * inv_d.z}, base, 4.0e-7f * magnitude};
}

// Entry distance of child k (its six bytes given as the words they sit in and the byte index), +inf when missed
ETX_HD float bvh8_slab(const Bvh8RayFrame& rf, uint32_t lox, uint32_t loy, uint32_t loz, uint32_t hix, uint32_t hiy, uint32_t hiz, uint32_t k, float tmin, float tmax) {
  const float tx0 = fmaf(bvh8_byte(lox, k), rf.step.x, rf.base.x), tx1 = fmaf(bvh8_byte(hix, k), rf.step.x, rf.base.x);
  const float ty0 = fmaf(bvh8_byte(loy, k), rf.step.y, rf.base.y), ty1 = fmaf(bvh8_byte(hiy, k), rf.step.y, rf.base.y);
  const float tz0 = fmaf(bvh8_byte(loz, k), rf.step.z, rf.base.z), tz1 = fmaf(bvh8_byte(hiz, k), rf.step.z, rf.base.z);
  const float t_enter = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
  const float t_exit = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
  return (t_enter <= t_exit * 1.0000004f + rf.slack) ? t_enter : kMaxFloat;
}

// Tests the eight children of a node against the ray segment [tmin, tmax]; returns the nearest hit child (kBvhEmptyChild: none) and
// pushes the other hit children. `kNearestFirst = false` (occlusion queries): the first hit child is taken, no distance is compared.
template <bool kNearestFirst, class Stack>
ETX_HD int32_t bvh8_visit(const Bvh8Words& w, const f3& ray_o, const f3& inv_d, float tmin, float tmax, const Stack& stack, uint32_t& sp) {
  const Bvh8RayFrame rf = bvh8_ray_frame(w.frame, ray_o, inv_d);
  float t_next = kMaxFloat;
  int32_t next = kBvhEmptyChild;
#define ETX_BVH8_CHILD(CHILD, LOX, LOY, LOZ, HIX, HIY, HIZ, K)                                 \
  {                                                                                            \
    const int32_t c_ = (CHILD);                                                                \
    const float t_ = bvh8_slab(rf, LOX, LOY, LOZ, HIX, HIY, HIZ, K, tmin, tmax);               \
    if ((c_ != kBvhEmptyChild) && (t_ < kMaxFloat)) {                                          \
      if (next == kBvhEmptyChild) {                                                            \
        next = c_, t_next = t_;                                                                \
      } else if (kNearestFirst && (t_ < t_next)) {                                             \
        stack.push(sp, next);                                                                  \
        next = c_, t_next = t_;                                                                \
      } else {                                                                                 \
        stack.push(sp, c_);                                                                    \
      }                                                                                        \
    }                                                                                          \
  }
  ETX_BVH8_CHILD(w.child_a.x, w.lo_xy.x, w.lo_xy.z, w.lo_z_hi_x.x, w.lo_z_hi_x.z, w.hi_yz.x, w.hi_yz.z, 0u)
  ETX_BVH8_CHILD(w.child_a.y, w.lo_xy.x, w.lo_xy.z, w.lo_z_hi_x.x, w.lo_z_hi_x.z, w.hi_yz.x, w.hi_yz.z, 1u)
  ETX_BVH8_CHILD(w.child_a.z, w.lo_xy.x, w.lo_xy.z, w.lo_z_hi_x.x, w.lo_z_hi_x.z, w.hi_yz.x, w.hi_yz.z, 2u)
  ETX_BVH8_CHILD(w.child_a.w, w.lo_xy.x, w.lo_xy.z, w.lo_z_hi_x.x, w.lo_z_hi_x.z, w.hi_yz.x, w.hi_yz.z, 3u)
  ETX_BVH8_CHILD(w.child_b.x, w.lo_xy.y, w.lo_xy.w, w.lo_z_hi_x.y, w.lo_z_hi_x.w, w.hi_yz.y, w.hi_yz.w, 0u)
  ETX_BVH8_CHILD(w.child_b.y, w.lo_xy.y, w.lo_xy.w, w.lo_z_hi_x.y, w.lo_z_hi_x.w, w.hi_yz.y, w.hi_yz.w, 1u)
  ETX_BVH8_CHILD(w.child_b.z, w.lo_xy.y, w.lo_xy.w, w.lo_z_hi_x.y, w.lo_z_hi_x.w, w.hi_yz.y, w.hi_yz.w, 2u)
  ETX_BVH8_CHILD(w.child_b.w, w.lo_xy.y, w.lo_xy.w, w.lo_z_hi_x.y, w.lo_z_hi_x.w, w.hi_yz.y, w.hi_yz.w, 3u)
#undef ETX_BVH8_CHILD
  return next;
}

#if defined(__HIPCC__)
// Where the kernels read the wide nodes: like BvhNodes, the first lds_count nodes from the workgroup's LDS copy
struct Bvh8Nodes {
  const uint4* global;
  const uint4* lds;
  uint32_t lds_count;
};

ETX_DEV Bvh8Nodes global_nodes8(const DScene& scene) {
  return {reinterpret_cast<const uint4*>(scene.bvh8_nodes), nullptr, 0u};
}

ETX_DEV Bvh8Nodes stage_nodes8(const DScene& scene, uint4* lds, uint32_t capacity_nodes) {
  const uint32_t count = min(scene.bvh8_node_count, capacity_nodes);
  const uint4* src = reinterpret_cast<const uint4*>(scene.bvh8_nodes);
  for (uint32_t i = threadIdx.x; i < count * 8u; i += blockDim.x)
    lds[i] = src[i];
  __syncthreads();
  return {src, lds, count};
}

ETX_DEV Bvh8Words bvh8_fetch(const Bvh8Nodes& nodes, int32_t cur) {
  return bvh8_load((uint32_t(cur) < nodes.lds_count) ? (nodes.lds + uint32_t(cur) * 8u) : (nodes.global + uint32_t(cur) * 8u));
}

// bvh_closest (dev_bvh.h) on the wide tree: same leaves, same filters, same result
template <class Tris, class Stack>
ETX_DEV Hit bvh_closest(const DScene& scene, const Bvh8Nodes& nodes, Tris tris, int32_t /* four-wide root, unused */, const Stack& stack, const RayQ& ray, uint32_t& alpha_seed,
  uint32_t* out_flags, uint32_t material_filter = kInvalid) {
  Hit best = {0.0f, 0.0f, ray.tmax, kInvalid};
  uint32_t best_flags = 0u;
  const f3 inv_d = {bvh8_reciprocal(ray.d.x), bvh8_reciprocal(ray.d.y), bvh8_reciprocal(ray.d.z)};
  uint32_t sp = 0;
  const int32_t kDone = kBvhEmptyChild;
  int32_t cur = (scene.bvh_tri_count == 0u) ? kDone : scene.bvh8_root;
  while (cur != kDone) {
    while ((cur >= 0) && (cur != kDone)) {
      const int32_t next = bvh8_visit<true>(bvh8_fetch(nodes, cur), ray.o, inv_d, ray.tmin, best.t, stack, sp);
      cur = (next != kDone) ? next : (sp ? stack.pop(sp) : kDone);
    }
    if (cur == kDone)
      break;
    const uint32_t leaf = uint32_t(~cur);
    const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
    for (uint32_t i = first; i < first + count; ++i) {
      const float4 v0 = tris[i].v0_index;
      const float4 e1 = tris[i].e1_flags;
      const float4 e2 = tris[i].e2_mat;
      float u, v, t;
      if (triangle_test(v0, e1, e2, ray, best.t, u, v, t) == false)
        continue;
      const uint32_t flags = __float_as_uint(e1.w);
      if (flags & kTriVoid)
        continue;
      if ((material_filter != kInvalid) && (__float_as_uint(e2.w) != material_filter))
        continue;
      const uint32_t tri_index = __float_as_uint(v0.w);
      if ((flags & kTriAlphaTested) && alpha_test_skips(scene, tri_index, __float_as_uint(e2.w), u, v, alpha_seed))
        continue;
      best = {u, v, t, tri_index};
      best_flags = flags;
    }
    cur = sp ? stack.pop(sp) : kDone;
  }
  if (out_flags)
    *out_flags = best_flags;
  return best;
}

// bvh_occluded (dev_bvh.h) on the wide tree
template <class Tris, class Stack>
ETX_DEV bool bvh_occluded(const DScene& scene, const Bvh8Nodes& nodes, Tris tris, int32_t /* four-wide root, unused */, const Stack& stack, const RayQ& ray, uint32_t& alpha_seed) {
  const f3 inv_d = {bvh8_reciprocal(ray.d.x), bvh8_reciprocal(ray.d.y), bvh8_reciprocal(ray.d.z)};
  uint32_t sp = 0;
  const int32_t kDone = kBvhEmptyChild;
  int32_t cur = (scene.bvh_tri_count == 0u) ? kDone : scene.bvh8_root;
  while (cur != kDone) {
    while ((cur >= 0) && (cur != kDone)) {
      const int32_t next = bvh8_visit<false>(bvh8_fetch(nodes, cur), ray.o, inv_d, ray.tmin, ray.tmax, stack, sp);
      cur = (next != kDone) ? next : (sp ? stack.pop(sp) : kDone);
    }
    if (cur == kDone)
      break;
    const uint32_t leaf = uint32_t(~cur);
    const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
    for (uint32_t i = first; i < first + count; ++i) {
      const float4 v0 = tris[i].v0_index;
      const float4 e1 = tris[i].e1_flags;
      const float4 e2 = tris[i].e2_mat;
      float u, v, t;
      if (triangle_test(v0, e1, e2, ray, ray.tmax, u, v, t) == false)
        continue;
      const uint32_t flags = __float_as_uint(e1.w);
      if (flags & kTriVoid)
        continue;
      if ((flags & kTriAlphaTested) && alpha_test_skips(scene, __float_as_uint(v0.w), __float_as_uint(e2.w), u, v, alpha_seed))
        continue;
      return true;
    }
    cur = sp ? stack.pop(sp) : kDone;
  }
  return false;
}
#endif  // __HIPCC__

}  // namespace etxd
