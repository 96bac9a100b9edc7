// dev_bvh.h - BVH2 traversal for gfx950: closest hit and transmittance ("shadow") queries.
//
// Replaces the reference's Embree calls (sources/etx/rt/rt.cxx:250-278 trace_with_function + the filter lambdas):
//   closest hit   = Raytracing::trace               rt.cxx:428-466 (skip Void, stochastic alpha, keep closest)
//   transmittance = Raytracing::trace_transmittance rt.cxx:468-579 (Boundary surfaces are transparent and switch
//                   the medium by the side of geo_n; anything else occludes; media attenuate the segments)
// Traversal state: a per-lane stack that lives in LDS (deep trees: its upper part in global memory), laid out [depth][lane] so a
// wavefront's pushes/pops hit 64 consecutive banks; nodes are 128-byte four-child packets (dev_scene.h Bvh4Node), the top of the tree staged in LDS
// by the traversal kernels (BvhNodes below), the rest read through L2.
#pragma once

#include "dev_scene.h"

namespace etxd {

constexpr uint32_t kStackDepth = 32;     // stack entries a lane keeps in LDS
constexpr uint32_t kMaxStackDepth = 512;  // deepest stack a tree may need (host bound over the tree: three pushes per BVH4 level = 170 levels); the entries above the LDS part
                                          // spill to a per-lane global area that is sized by the uploaded tree's OWN bound (host_api.cpp allocate_pipeline). Until round 5 the
                                          // limit was 64 and deeper trees were refused; Embree has no such limit (rt.cxx:66-88)
constexpr uint32_t kFlatSweepMaxTriangles = 64;  // scenes up to this size are swept linearly (all lanes, same triangle)

// The stack of the two traversal kernels when the tree's bound fits the LDS part (trees up to ~40 000 triangles): no checks.
struct FastLaneStack {
  int32_t* base;    // LDS, this lane's slot of level 0
  uint32_t stride;  // lanes per level (= block size)
  ETX_DEV void push(uint32_t& sp, int32_t v) const {
    base[sp * stride] = v;
    sp += 1u;
  }
  ETX_DEV int32_t pop(uint32_t& sp) const {
    sp -= 1u;
    return base[sp * stride];
  }
};

// The general stack: kStackDepth entries in LDS, the rest in global memory (DScene::stack_spill, allocated per device lane when the
// uploaded tree's bound exceeds kStackDepth; laid out [level][lane] like the LDS part). The bound is a worst case over the tree -
// three pushed children on every level of the deepest path: 43 entries for a million triangles -, real rays stay below 20, so the
// spill is a guarantee rather than traffic.
struct LaneStack {
  int32_t* base;
  uint32_t stride;
  int32_t* spill;
  uint32_t spill_stride;
  ETX_DEV void push(uint32_t& sp, int32_t v) const {
    if (sp < kStackDepth)
      base[sp * stride] = v;
    else
      spill[(sp - kStackDepth) * spill_stride] = v;
    sp += 1u;
  }
  ETX_DEV int32_t pop(uint32_t& sp) const {
    sp -= 1u;
    return (sp < kStackDepth) ? base[sp * stride] : spill[(sp - kStackDepth) * spill_stride];
  }
};

// The same with kShortStackDepth entries in LDS: kernels that are short of LDS rather than of registers (16 KB of stacks per workgroup
// instead of 32). Real rays stay below 20 entries, so a few per cent of the pushes go to the spill rows (level - kShortStackDepth).
constexpr uint32_t kShortStackDepth = 16;
struct ShortLaneStack {
  int32_t* base;
  uint32_t stride;
  int32_t* spill;
  uint32_t spill_stride;
  ETX_DEV void push(uint32_t& sp, int32_t v) const {
    if (sp < kShortStackDepth)
      base[sp * stride] = v;
    else
      spill[(sp - kShortStackDepth) * spill_stride] = v;
    sp += 1u;
  }
  ETX_DEV int32_t pop(uint32_t& sp) const {
    sp -= 1u;
    return (sp < kShortStackDepth) ? base[sp * stride] : spill[(sp - kShortStackDepth) * spill_stride];
  }
};

ETX_DEV ShortLaneStack short_lane_stack(const DScene& scene, int32_t* lds_slot, uint32_t stride) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t* spill = ((scene.stack_spill != nullptr) && (lane < scene.stack_spill_lanes)) ? (scene.stack_spill + lane) : nullptr;
  return {lds_slot, stride, spill, scene.stack_spill_lanes};
}

// `lds_slot`: this lane's slot of level 0 in the workgroup's stack array, `stride`: lanes per level
ETX_DEV LaneStack lane_stack(const DScene& scene, int32_t* lds_slot, uint32_t stride) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t* spill = ((scene.stack_spill != nullptr) && (lane < scene.stack_spill_lanes)) ? (scene.stack_spill + lane) : nullptr;
  return {lds_slot, stride, spill, scene.stack_spill_lanes};
}

struct Hit {
  float u, v, t;
  uint32_t tri;  // kInvalid = miss
};

struct RayQ {
  f3 o;
  float tmin;
  f3 d;
  float tmax;
};

ETX_DEV float alpha_random(uint32_t& alpha_seed) {
  // one uniform per alpha-tested candidate (scene_bsdf.hxx:143 draws smp.next() per candidate);
  // the stream is private to the ray so the traversal kernels never write path state.
  Sampler s;
  s.seed = alpha_seed;
  float r = s.next();
  alpha_seed = s.seed;
  return r;
}

// scene_bsdf.hxx:128-144 alpha_test_pass: true = the candidate is skipped
ETX_DEV bool alpha_test_skips(const DScene& scene, uint32_t tri_index, uint32_t material_index, float u, float v, uint32_t& alpha_seed) {
  const etx_abi_material& mat = scene.materials[material_index];
  float alpha = mat.opacity;
  if (mat.scattering.image_index != kInvalid) {
    const DImage& img = scene.images[mat.scattering.image_index];
    if (img.options & ETX_IMAGE_HAS_ALPHA) {
      f2 uv = lerp_uv(scene, scene.triangles[tri_index], barycentrics(u, v));
      ImageGather g = image_gather(img, uv);
      alpha *= g.p00.w + g.p01.w + g.p10.w + g.p11.w;
    }
  }
  return alpha <= alpha_random(alpha_seed);
}

// Slab test of one child box; returns entry distance or +inf when missed.
ETX_DEV float slab(const f3& lo, const f3& hi, const f3& o, const f3& inv_d, float tmin, float tmax) {
  float tx0 = (lo.x - o.x) * inv_d.x, tx1 = (hi.x - o.x) * inv_d.x;
  float ty0 = (lo.y - o.y) * inv_d.y, ty1 = (hi.y - o.y) * inv_d.y;
  float tz0 = (lo.z - o.z) * inv_d.z, tz1 = (hi.z - o.z) * inv_d.z;
  float t_enter = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
  float t_exit = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
  return (t_enter <= t_exit * 1.0000004f) ? t_enter : kMaxFloat;
}

enum : uint32_t {
  kQueryClosest = 0,  // every non-void, alpha-passing triangle is a candidate
  kQueryShadowAny = 1 // as closest, used by the transmittance walk (the caller inspects the material class)
};

// One Moeller-Trumbore test; u,v are the barycentrics of vertices 1 and 2 (Embree convention, rt.cxx:352-353).
// Returns true when the triangle is a candidate closer than best.t (Void / alpha filter not applied yet).
ETX_DEV bool triangle_test(const float4& v0, const float4& e1, const float4& e2, const RayQ& ray, float t_limit, float& out_u, float& out_v, float& out_t) {
  const f3 E1 = {e1.x, e1.y, e1.z}, E2 = {e2.x, e2.y, e2.z};
  const f3 p = cross(ray.d, E2);
  const float det = dot(E1, p);
  const float inv_det = __builtin_amdgcn_rcpf(det);  // v_rcp_f32 (1 ulp): t/u/v stay within 1e-6 relative of the IEEE division
  const f3 s = ray.o - f3{v0.x, v0.y, v0.z};
  const float u = dot(s, p) * inv_det;
  const f3 q = cross(s, E1);
  const float v = dot(ray.d, q) * inv_det;
  const float t = dot(E2, q) * inv_det;
  out_u = u, out_v = v, out_t = t;
  // det == 0 gives inf/nan which fail the comparisons below
  return (det != 0.0f) && (u >= 0.0f) && (u <= 1.0f) && (v >= 0.0f) && (u + v <= 1.0f) && (t >= ray.tmin) && (t <= t_limit);
}

// Linear sweep for tiny scenes (Cornell: 32-44 triangles): the loop index is wave uniform, so the triangle records
// are fetched once per wave through the scalar cache and every lane runs the same instruction stream - no stack, no
// divergence, no dependent node fetches. ~45 VALU per triangle and ray.
// The triangle table is read through the constant address space: the kernels that sweep also store hits, so the
// compiler has to assume ordinary global loads are clobbered and emits per-lane flat loads followed by a full
// s_waitcnt for every triangle (measured: 141 L1 accesses per ray, 28 % VALU). Constant-address-space loads with a
// wave-uniform address become s_load_dwordx4 into SGPRs, can be issued several triangles ahead and cost no VGPRs.
typedef const __attribute__((address_space(4))) float* ConstantFloats;

struct FlatRow {  // one FlatPrim as it sits in SGPRs
  float4 plane, row_a, row_b;
  uint32_t flags, material;
  uint32_t medium_against, medium_along;
};

ETX_DEV FlatRow load_flat_prim(ConstantFloats table, uint32_t i) {
  ConstantFloats t = table + i * 16u;
  return {make_float4(t[0], t[1], t[2], t[3]), make_float4(t[4], t[5], t[6], t[7]), make_float4(t[8], t[9], t[10], t[11]), __float_as_uint(t[12]), __float_as_uint(t[13]),
    __float_as_uint(t[14]), __float_as_uint(t[15])};
}

// Plane hit, then parallelogram coordinates of the hit point. A triangle needs a + b <= 1, a parallelogram a <= 1 and
// b <= 1; the primitive kind is wave-uniform, so the two upper bounds become e = 1 - b - (quad ? 0 : a) >= 0 and
// f = 1 - (quad ? a : 0) >= 0 and the four lower bounds fold into one v_min3 + v_min. A ray parallel to the plane gives
// t = inf / nan, which fails the two explicit comparisons on t before the (NaN-dropping) minima matter.
ETX_DEV bool flat_prim_test(const FlatRow& prim, const RayQ& ray, float t_limit, float& out_a, float& out_b, float& out_t) {
  const float den = prim.plane.x * ray.d.x + prim.plane.y * ray.d.y + prim.plane.z * ray.d.z;
  const float num = prim.plane.x * ray.o.x + prim.plane.y * ray.o.y + prim.plane.z * ray.o.z + prim.plane.w;
  const float t = -num * __builtin_amdgcn_rcpf(den);
  const f3 x = ray.o + ray.d * t;
  const float a = prim.row_a.x * x.x + prim.row_a.y * x.y + prim.row_a.z * x.z + prim.row_a.w;
  const float b = prim.row_b.x * x.x + prim.row_b.y * x.y + prim.row_b.z * x.z + prim.row_b.w;
  const float quad = (prim.flags & kTriQuad) ? 1.0f : 0.0f;  // scalar
  const float e = (1.0f - b) - (1.0f - quad) * a;
  const float f = 1.0f - quad * a;
  const float inside = fminf(fminf(fminf(a, b), e), f);
  out_a = a, out_b = b, out_t = t;
  return (t >= ray.tmin) && (t <= t_limit) && (inside >= 0.0f);
}

// (primitive, a, b) -> (triangle, u, v)
ETX_DEV Hit flat_resolve(const DScene& scene, uint32_t prim, float a, float b, float t) {
  const FlatPrimInfo& info = scene.flat_info[prim];
  const bool second = (info.tri_b != kInvalid) && (a + b > 1.0f);
  const float* cu = second ? info.ub : info.ua;
  const float* cv = second ? info.vb : info.va;
  return {cu[0] + cu[1] * a + cu[2] * b, cv[0] + cv[1] * a + cv[2] * b, t, second ? info.tri_b : info.tri_a};
}

template <class Tris>
ETX_DEV Hit bvh_flat_closest(const DScene& scene, Tris, const RayQ& ray, uint32_t& alpha_seed, uint32_t* out_flags, uint32_t material_filter = kInvalid) {
  float best_a = 0.0f, best_b = 0.0f, best_t = ray.tmax;
  uint32_t best_prim = kInvalid;
  uint32_t best_flags = 0u;
  const uint32_t count = scene.flat_prim_count;
  ConstantFloats table = (ConstantFloats)(const void*)(scene.flat_prims);
#pragma unroll 4
  for (uint32_t i = 0; i < count; ++i) {
    const FlatRow prim = load_flat_prim(table, i);
    const uint32_t flags = prim.flags;
    float a, b, t;
    if (flat_prim_test(prim, ray, best_t, a, b, t) == false)
      continue;
    if (flags & kTriVoid)
      continue;
    if ((material_filter != kInvalid) && (prim.material != material_filter))  // trace_material, rt.cxx:342-345
      continue;
    // alpha-tested triangles are never merged into parallelograms: (a, b) are the triangle's own barycentrics
    if ((flags & kTriAlphaTested) && alpha_test_skips(scene, scene.flat_info[i].tri_a, prim.material, a, b, alpha_seed))
      continue;
    best_a = a, best_b = b, best_t = t, best_prim = i;
    best_flags = flags;
  }
  if (out_flags)
    *out_flags = best_flags;
  if (best_prim == kInvalid)
    return {0.0f, 0.0f, ray.tmax, kInvalid};
  return flat_resolve(scene, best_prim, best_a, best_b, best_t);
}

// Where the traversal reads its nodes: the first `lds_count` nodes (breadth-first numbering = the top of the tree) from the
// workgroup's LDS copy when the kernel staged one (north_star: "LDS-staged BVH node packets"), the rest from global
// memory (L2). 8 float4 per node.
struct BvhNodes {
  const float4* global;
  const float4* lds;
  uint32_t lds_count;
};

ETX_DEV BvhNodes global_nodes(const DScene& scene) {
  return {reinterpret_cast<const float4*>(scene.bvh_nodes), nullptr, 0u};
}

// Cooperative copy of the top of the tree into LDS (call from workgroup-uniform control flow, ends with a barrier).
ETX_DEV BvhNodes stage_nodes(const DScene& scene, float4* lds, uint32_t capacity_nodes) {
  const uint32_t count = min(scene.bvh_node_count, capacity_nodes);
  const float4* src = reinterpret_cast<const float4*>(scene.bvh_nodes);
  for (uint32_t i = threadIdx.x; i < count * 8u; i += blockDim.x)
    lds[i] = src[i];
  __syncthreads();
  return {src, lds, count};
}

ETX_DEV void sort_pair(float& ta, int32_t& ca, float& tb, int32_t& cb) {  // compare-exchange, ascending t
  const bool swap = tb < ta;
  const float t0 = swap ? tb : ta, t1 = swap ? ta : tb;
  const int32_t c0 = swap ? cb : ca, c1 = swap ? ca : cb;
  ta = t0, tb = t1, ca = c0, cb = c1;
}

// Closest accepted hit in [tmin, tmax]: BVH4, per-lane stack in LDS, near child first.
template <class Tris, class Stack>
ETX_DEV Hit bvh_closest(const DScene& scene, const BvhNodes& nodes, Tris tris, int32_t root, const Stack& stack, const RayQ& ray, uint32_t& alpha_seed, uint32_t* out_flags,
  uint32_t material_filter = kInvalid) {
  if (scene.bvh_flat)
    return bvh_flat_closest(scene, tris, ray, alpha_seed, out_flags, material_filter);
  Hit best = {0.0f, 0.0f, ray.tmax, kInvalid};
  uint32_t best_flags = 0u;
  const f3 inv_d = {__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z)};
  uint32_t sp = 0;
  int32_t cur = root;
  const int32_t kDone = kBvhEmptyChild;
  if (scene.bvh_tri_count == 0u)
    cur = kDone;
  // "while-while" (Aila & Laine): all lanes of the wave walk inner nodes until each of them stands on a leaf (or is done),
  // then all of them intersect their leaves - the leaf code runs with most lanes active instead of being interleaved with
  // the node code of the other lanes.
  while (cur != kDone) {
    while ((cur >= 0) && (cur != kDone)) {
      float4 lox, loy, loz, hix, hiy, hiz;
      int4 children;
      if (uint32_t(cur) < nodes.lds_count) {
        const float4* n = nodes.lds + uint32_t(cur) * 8u;
        lox = n[0], loy = n[1], loz = n[2], hix = n[3], hiy = n[4], hiz = n[5];
        const float4 c = n[6];
        children = make_int4(__float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w));
      } else {
        const float4* n = nodes.global + uint32_t(cur) * 8u;
        lox = n[0], loy = n[1], loz = n[2], hix = n[3], hiy = n[4], hiz = n[5];
        const float4 c = n[6];
        children = make_int4(__float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w));
      }
      // four slab tests; an unused slot holds lo = +inf, hi = -inf and is masked by its child code
      float t0 = slab(f3{lox.x, loy.x, loz.x}, f3{hix.x, hiy.x, hiz.x}, ray.o, inv_d, ray.tmin, best.t);
      float t1 = slab(f3{lox.y, loy.y, loz.y}, f3{hix.y, hiy.y, hiz.y}, ray.o, inv_d, ray.tmin, best.t);
      float t2 = slab(f3{lox.z, loy.z, loz.z}, f3{hix.z, hiy.z, hiz.z}, ray.o, inv_d, ray.tmin, best.t);
      float t3 = slab(f3{lox.w, loy.w, loz.w}, f3{hix.w, hiy.w, hiz.w}, ray.o, inv_d, ray.tmin, best.t);
      int32_t c0 = children.x, c1 = children.y, c2 = children.z, c3 = children.w;
      t0 = (c0 == kBvhEmptyChild) ? kMaxFloat : t0;
      t1 = (c1 == kBvhEmptyChild) ? kMaxFloat : t1;
      t2 = (c2 == kBvhEmptyChild) ? kMaxFloat : t2;
      t3 = (c3 == kBvhEmptyChild) ? kMaxFloat : t3;
      // sorting network of four (t, child) pairs; misses (t = max) end up last
      sort_pair(t0, c0, t1, c1);
      sort_pair(t2, c2, t3, c3);
      sort_pair(t0, c0, t2, c2);
      sort_pair(t1, c1, t3, c3);
      sort_pair(t1, c1, t2, c2);
      if (t0 == kMaxFloat) {
        cur = sp ? stack.pop(sp) : kDone;
      } else {  // farthest first, so that the nearest of the rest is popped first
        if (t3 < kMaxFloat)
          stack.push(sp, c3);
        if (t2 < kMaxFloat)
          stack.push(sp, c2);
        if (t1 < kMaxFloat)
          stack.push(sp, c1);
        cur = c0;
      }
    }
    if (cur == kDone)
      break;
    {
      uint32_t leaf = uint32_t(~cur);
      uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
      for (uint32_t i = first; i < first + count; ++i) {
        const float4 v0 = tris[i].v0_index;
        const float4 e1 = tris[i].e1_flags;
        const float4 e2 = tris[i].e2_mat;
        float u, v, t;
        if (triangle_test(v0, e1, e2, ray, best.t, u, v, t) == false)
          continue;
        uint32_t flags = __float_as_uint(e1.w);
        if (flags & kTriVoid)
          continue;
        if ((material_filter != kInvalid) && (__float_as_uint(e2.w) != material_filter))
          continue;
        uint32_t tri_index = __float_as_uint(v0.w);
        if ((flags & kTriAlphaTested) && alpha_test_skips(scene, tri_index, __float_as_uint(e2.w), u, v, alpha_seed))
          continue;
        best = {u, v, t, tri_index};
        best_flags = flags;
      }
      cur = sp ? stack.pop(sp) : kDone;
    }
  }
  if (out_flags)
    *out_flags = best_flags;
  return best;
}


// Any accepted hit in [tmin, tmax]? The occlusion test of a scene WITHOUT Boundary materials (Raytracing::trace_transmittance,
// rt.cxx:488-516: the first candidate that is neither Void nor alpha-skipped ends the query). No order among the children of a node
// is needed and nothing is tracked: a traversal that stops at the first hit instead of shrinking the interval around the nearest one.
template <class Tris, class Stack>
ETX_DEV bool bvh_occluded(const DScene& scene, const BvhNodes& nodes, Tris tris, int32_t root, const Stack& stack, const RayQ& ray, uint32_t& alpha_seed) {
  const f3 inv_d = {__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z)};
  uint32_t sp = 0;
  int32_t cur = root;
  const int32_t kDone = kBvhEmptyChild;
  if (scene.bvh_tri_count == 0u)
    cur = kDone;
  while (cur != kDone) {
    while ((cur >= 0) && (cur != kDone)) {
      float4 lox, loy, loz, hix, hiy, hiz;
      int4 children;
      if (uint32_t(cur) < nodes.lds_count) {
        const float4* n = nodes.lds + uint32_t(cur) * 8u;
        lox = n[0], loy = n[1], loz = n[2], hix = n[3], hiy = n[4], hiz = n[5];
        const float4 c = n[6];
        children = make_int4(__float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w));
      } else {
        const float4* n = nodes.global + uint32_t(cur) * 8u;
        lox = n[0], loy = n[1], loz = n[2], hix = n[3], hiy = n[4], hiz = n[5];
        const float4 c = n[6];
        children = make_int4(__float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w));
      }
      const float t0 = slab(f3{lox.x, loy.x, loz.x}, f3{hix.x, hiy.x, hiz.x}, ray.o, inv_d, ray.tmin, ray.tmax);
      const float t1 = slab(f3{lox.y, loy.y, loz.y}, f3{hix.y, hiy.y, hiz.y}, ray.o, inv_d, ray.tmin, ray.tmax);
      const float t2 = slab(f3{lox.z, loy.z, loz.z}, f3{hix.z, hiy.z, hiz.z}, ray.o, inv_d, ray.tmin, ray.tmax);
      const float t3 = slab(f3{lox.w, loy.w, loz.w}, f3{hix.w, hiy.w, hiz.w}, ray.o, inv_d, ray.tmin, ray.tmax);
      const bool h0 = (children.x != kBvhEmptyChild) && (t0 < kMaxFloat), h1 = (children.y != kBvhEmptyChild) && (t1 < kMaxFloat);
      const bool h2 = (children.z != kBvhEmptyChild) && (t2 < kMaxFloat), h3 = (children.w != kBvhEmptyChild) && (t3 < kMaxFloat);
      int32_t next = kDone;
      if (h3)
        next = children.w;
      if (h2) {
        if (next != kDone)
          stack.push(sp, next);
        next = children.z;
      }
      if (h1) {
        if (next != kDone)
          stack.push(sp, next);
        next = children.y;
      }
      if (h0) {
        if (next != kDone)
          stack.push(sp, next);
        next = children.x;
      }
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

// scene_medium.hxx:187-193 (homogeneous branch): exp(-sigma_t * distance)
// medium_transmittance, scene_medium.hxx:191-239: homogeneous exp(-sigma_t d); heterogeneous ratio tracking against the
// majorant with Russian roulette below 0.1 (draws from `smp`: the per-request stream of the shadow kernel).
ETX_DEV f3 medium_transmittance(const DScene& scene, const DMedium& m, float wavelength, Sampler& smp, const f3& pos, const f3& direction, float distance) {
  const float4 row = reinterpret_cast<const float4*>(&m)[2];  // extinction (RGB mode), class: one load (DMedium, dev_scene.h)
  if (__float_as_uint(row.w) == 0u) {
    f3 ext = {row.x, row.y, row.z};
    if (scene.spectral != 0u) {
      f3 absorption, scattering;
      medium_coefficients(scene, m, wavelength, absorption, scattering);
      ext = absorption + scattering;
    }
    return {expf(-ext.x * distance), expf(-ext.y * distance), expf(-ext.z * distance)};
  }
  if (m.max_sigma <= 0.0f)
    return mk3(1.0f);
  f3 medium_pos, medium_dir;
  float t_min = 0.0f, t_max = 0.0f;
  if (medium_intersects_bounds(m, pos, direction, distance, medium_pos, medium_dir, t_min, t_max) == false)
    return mk3(1.0f);
  const float rr_threshold = 0.1f;
  float transmittance = 1.0f;
  float t = t_min;
  while (true) {
    t -= logf(1.0f - smp.next()) / m.max_sigma;
    if (t >= t_max)
      break;
    const float density_value = medium_sample_density(m, medium_pos + medium_dir * t);
    transmittance *= fmaxf(0.0f, 1.0f - density_value);
    if (transmittance < rr_threshold) {
      const float q = fmaxf(0.05f, 1.0f - transmittance);
      if (smp.next() < q)
        return mk3(0.0f);
      transmittance /= (1.0f - q);
    }
  }
  return mk3(transmittance);
}

// Flat-sweep transmittance for tiny scenes: ONE pass over all triangles finds (a) any occluder and (b) up to four
// Boundary crossings kept sorted by t in registers (rt.cxx:488-516 collects up to 63 and sorts, :518-578 walks the
// media); more than four crossings fall back to the restart walk below. Returns false when the fallback is needed.
template <class Tris>
ETX_DEV bool flat_transmittance(const DScene& scene, Tris tris, const f3& p0, const f3& direction, float t_max, uint32_t medium_index, float wavelength, uint32_t& alpha_seed, f3& result) {
  const RayQ ray = {p0, kRayEpsilon, direction, t_max};
  float bt0 = kMaxFloat, bt1 = kMaxFloat, bt2 = kMaxFloat, bt3 = kMaxFloat;
  uint32_t bi0 = kInvalid, bi1 = kInvalid, bi2 = kInvalid, bi3 = kInvalid;
  uint32_t crossings = 0;
  bool occluded = false;
  const uint32_t count = scene.flat_prim_count;
  ConstantFloats table = (ConstantFloats)(const void*)(scene.flat_prims);
#pragma unroll 4
  for (uint32_t i = 0; i < count; ++i) {
    const FlatRow tri = load_flat_prim(table, i);
    const uint32_t flags = tri.flags;
    float u, v, t;
    if (flat_prim_test(tri, ray, t_max, u, v, t) == false)
      continue;
    if (flags & kTriVoid)
      continue;
    if ((flags & kTriAlphaTested) && alpha_test_skips(scene, scene.flat_info[i].tri_a, tri.material, u, v, alpha_seed))
      continue;
    if ((flags & kTriBoundary) == 0u) {
      occluded = true;
      continue;
    }
    crossings++;
    // insertion into the sorted 4-slot list (compare-exchange chain, all in registers). What is kept of a crossing is the medium BEYOND it:
    // the row holds both candidates in scalar registers and the side is the sign of N . d (FlatPrim::medium_against / _along) - walking the
    // media afterwards gathers nothing (it read primitive info -> triangle -> material per crossing: three dependent loads)
    float ct = t;
    uint32_t ci = ((tri.plane.x * direction.x + tri.plane.y * direction.y + tri.plane.z * direction.z) < 0.0f) ? tri.medium_against : tri.medium_along;
    if (ct < bt0) { float tt = bt0; uint32_t ti = bi0; bt0 = ct, bi0 = ci, ct = tt, ci = ti; }
    if (ct < bt1) { float tt = bt1; uint32_t ti = bi1; bt1 = ct, bi1 = ci, ct = tt, ci = ti; }
    if (ct < bt2) { float tt = bt2; uint32_t ti = bi2; bt2 = ct, bi2 = ci, ct = tt, ci = ti; }
    if (ct < bt3) { bt3 = ct, bi3 = ci; }
  }
  if (occluded) {
    result = mk3(0.0f);
    return true;
  }
  if (crossings > 4u)
    return false;
  result = mk3(1.0f);
  float current_t = 0.0f;
  Sampler medium_rng;  // heterogeneous media draw from the segment's own stream
  medium_rng.seed = alpha_seed ^ 0x6d656469u, medium_rng.fixed_u = medium_rng.fixed_v = medium_rng.fixed_w = 0.0f;
  uint32_t medium = medium_index;
  const float bts[4] = {bt0, bt1, bt2, bt3};
  const uint32_t bis[4] = {bi0, bi1, bi2, bi3};
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    if (k < crossings) {
      if (medium != kInvalid)
        result *= medium_transmittance(scene, scene.mediums[medium], wavelength, medium_rng, p0 + direction * current_t, direction, fmaxf(0.0f, bts[k] - current_t));
      medium = bis[k];
      current_t = bts[k];
    }
  }
  if (medium != kInvalid)
    result *= medium_transmittance(scene, scene.mediums[medium], wavelength, medium_rng, p0 + direction * current_t, direction, fmaxf(0.0f, t_max - current_t));
  return true;
}

// The same query on a tree scene that holds no Class::Boundary material and no density grid (DScene::boundary_materials,
// heterogeneous_mediums): the segment is occluded or it is not, and what it crosses is the homogeneous medium it started in -
// one any-hit traversal and one exp, a third fewer registers than the general function (k_trace_shadow<false, kDeep, true>).
template <class Nodes, class Tris, class Stack>
ETX_DEV f3 bvh_transmittance_opaque(const DScene& scene, const Nodes& nodes, Tris tris, int32_t root, const Stack& stack, const f3& p0, const f3& p1, uint32_t medium_index,
  float wavelength, uint32_t& alpha_seed) {
  f3 direction = p1 - p0;
  float t_max = dot(direction, direction);
  if (t_max <= kRayEpsilon)
    return mk3(1.0f);
  t_max = sqrtf(t_max);
  direction = direction / t_max;
  t_max -= fmaxf(kRayEpsilon, t_max * kRayEpsilon);
  if (bvh_occluded(scene, nodes, tris, root, stack, RayQ{p0, kRayEpsilon, direction, t_max}, alpha_seed))
    return mk3(0.0f);
  if (medium_index == kInvalid)
    return mk3(1.0f);
  const float4 row = reinterpret_cast<const float4*>(&scene.mediums[medium_index])[2];  // extinction (RGB mode)
  f3 ext = {row.x, row.y, row.z};
  if (scene.spectral != 0u) {
    f3 absorption, scattering;
    medium_coefficients(scene, scene.mediums[medium_index], wavelength, absorption, scattering);
    ext = absorption + scattering;
  }
  return {expf(-ext.x * t_max), expf(-ext.y * t_max), expf(-ext.z * t_max)};
}

// Transmittance between p0 and p1 starting in `medium_index` (rt.cxx:468-579).
// The reference collects up to 63 Boundary hits in one traversal and sorts them; here the boundaries are visited in
// order by restarting the closest-hit search behind each one (same products, no per-lane hit buffer).
// rays_traced counts the traversals (statistics).
template <class Tris, class Stack>
ETX_DEV f3 bvh_transmittance(const DScene& scene, const BvhNodes& nodes, Tris tris, int32_t root, const Stack& stack, const f3& p0, const f3& p1, uint32_t medium_index,
  float wavelength, uint32_t& alpha_seed) {
  f3 direction = p1 - p0;
  float t_max = dot(direction, direction);
  if (t_max <= kRayEpsilon)
    return mk3(1.0f);
  t_max = sqrtf(t_max);
  direction = direction / t_max;
  t_max -= fmaxf(kRayEpsilon, t_max * kRayEpsilon);

  f3 result = mk3(1.0f);
  if (scene.bvh_flat && flat_transmittance(scene, tris, p0, direction, t_max, medium_index, wavelength, alpha_seed, result))
    return result;
  result = mk3(1.0f);
  float current_t = 0.0f;
  Sampler medium_rng;  // heterogeneous media draw from the segment's own stream
  medium_rng.seed = alpha_seed ^ 0x6d656469u, medium_rng.fixed_u = medium_rng.fixed_v = medium_rng.fixed_w = 0.0f;
  float t_min = kRayEpsilon;
  if ((scene.bvh_flat == 0u) && (scene.boundary_materials == 0u)) {
    // no medium boundary anywhere: the segment is occluded or it is not, and what it crosses is the medium it started in
    if (bvh_occluded(scene, nodes, tris, root, stack, RayQ{p0, t_min, direction, t_max}, alpha_seed))
      return mk3(0.0f);
    return (medium_index != kInvalid) ? medium_transmittance(scene, scene.mediums[medium_index], wavelength, medium_rng, p0, direction, t_max) : mk3(1.0f);
  }
  uint32_t medium = medium_index;
  for (uint32_t crossings = 0; crossings < 64u; ++crossings) {
    uint32_t flags = 0u;
    Hit h = bvh_closest(scene, nodes, tris, root, stack, RayQ{p0, t_min, direction, t_max}, alpha_seed, &flags);
    bool found = h.tri != kInvalid;
    if (found && ((flags & kTriBoundary) == 0u))
      return mk3(0.0f);
    float seg_end = found ? h.t : t_max;
    if (medium != kInvalid) {
      float dt = fmaxf(0.0f, seg_end - current_t);
      const DMedium& m = scene.mediums[medium];
      // heterogeneous media (ratio tracking, scene_medium.hxx:195-232) are rejected at upload for now
      result *= medium_transmittance(scene, m, wavelength, medium_rng, p0 + direction * current_t, direction, dt);
    }
    if (found == false)
      return result;
    const etx_abi_triangle& tri = scene.triangles[h.tri];
    const etx_abi_material& mat = scene.materials[tri.material_index];
    bool entering = dot(ld3(tri.geo_n), direction) < 0.0f;
    medium = entering ? mat.int_medium : mat.ext_medium;
    current_t = h.t;
    t_min = __uint_as_float(__float_as_uint(h.t) + 1u);  // strictly behind this boundary
    if (t_min > t_max)
      return result;
  }
  return mk3(0.0f);  // rt.cxx:503: more boundaries than the buffer holds counts as occlusion
}

}  // namespace etxd
