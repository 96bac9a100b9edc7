// dev_lbvh.h - the per-element steps of the device BVH build, written once for the kernels (kernels_bvh_build.hip,
// one thread per element) and for the host emulation of the same build (host_scene.cpp build_lbvh_host: the CPU tests check the
// tree the device WILL build - invariants, stack bound, ray for ray against the SAH tree - without a GPU).
//
// The build replaces what Raytracing::commit_changes hands to Embree (sources/etx/rt/rt.cxx:58-88) when the geometry changed:
//   1. 63-bit Morton key of every triangle's centroid inside the scene's bounding cube, radix sort of (key, triangle)
//   2. the binary radix tree over the sorted keys (Karras 2012: every inner node finds its range and split on its own)
//   3. boxes of the radix tree's nodes, bottom-up (the second thread to reach a node joins its children's boxes)
//   4. collapse to the four-wide breadth-first nodes the traversal reads (dev_scene.h Bvh4Node): level by level, a node adopts
//      the children of its child with the largest surface area until it has four (what the host builder's collapse does); a range
//      of <= kLbvhLeafMax triangles becomes a leaf when the surface area heuristic prefers that to its split
//   5. boxes bottom-up by level from the triangles' vertices (the refit of etx_hip_update_scene), stack bound by level
// A linear BVH traverses slower than the host's binned-SAH tree (measured: DESIGN.md); it is the builder for geometry that changes.
#pragma once

#include "dev_scene.h"
#include "dev_math.h"

namespace etxd {

constexpr uint32_t kLbvhLeafMax = 4;  // triangles per leaf (the leaf reference holds count - 1 in 3 bits)

ETX_HD uint32_t lbvh_float_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
#endif
}

ETX_HD float lbvh_bits_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
#endif
}

ETX_HD uint64_t lbvh_spread21(uint32_t v) {  // 21 bits -> every third bit of 63
  uint64_t x = v & 0x1fffffull;
  x = (x | (x << 32)) & 0x1f00000000ffffull;
  x = (x | (x << 16)) & 0x1f0000ff0000ffull;
  x = (x | (x << 8)) & 0x100f00f00f00f00full;
  x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

// centroid of triangle `ti` in the cube [cube_min, cube_min + cube_extent]^3 -> Morton key
ETX_HD uint64_t lbvh_morton_key(const DScene& scene, uint32_t ti, const f3& cube_min, float inv_extent) {
  const etx_abi_triangle& t = scene.triangles[ti];
  f3 lo = mk3(kMaxFloat), hi = mk3(-kMaxFloat);
  for (uint32_t c = 0; c < 3u; ++c) {
    const etx_abi_vertex& v = scene.vertices[t.i[c]];
    const f3 p = {v.pos.x, v.pos.y, v.pos.z};
    lo = fmin3(lo, p), hi = fmax3(hi, p);
  }
  const f3 unit = ((lo + hi) * 0.5f - cube_min) * inv_extent;
  auto quantize = [](float u) {
    const float q = u * 2097152.0f;  // 2^21
    return uint32_t((q < 0.0f) ? 0.0f : ((q > 2097151.0f) ? 2097151.0f : q));  // (a NaN centroid lands in cell 0)
  };
  return (lbvh_spread21(quantize(unit.x)) << 2) | (lbvh_spread21(quantize(unit.y)) << 1) | lbvh_spread21(quantize(unit.z));
}

// length of the common prefix of keys i and j, ties broken by the position (Karras 2012, section 4); -1 outside the array
ETX_HD int lbvh_delta(const uint64_t* keys, int n, int i, int j) {
  if ((j < 0) || (j >= n))
    return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a == b)
    return 64 + __builtin_clz(uint32_t(i) ^ uint32_t(j));
  return __builtin_clzll(a ^ b);
}

// Inner node i of the binary radix tree over n sorted keys (0 <= i < n - 1): it covers the sorted positions [first, last], its left
// child [first, split], its right child [split + 1, last]. A child that covers more than one position is inner node `split`
// (left) or `split + 1` (right).
struct LbvhNode {
  uint32_t first, last, split, pad;
};

ETX_HD LbvhNode lbvh_node(const uint64_t* keys, int n, int i) {
  const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int delta_min = lbvh_delta(keys, n, i, i - d);
  int l_max = 2;
  while (lbvh_delta(keys, n, i, i + l_max * d) > delta_min)
    l_max *= 2;
  int l = 0;
  for (int t = l_max / 2; t >= 1; t /= 2)
    if (lbvh_delta(keys, n, i, i + (l + t) * d) > delta_min)
      l += t;
  const int j = i + l * d;
  const int delta_node = lbvh_delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {  // ceil(l / 2), ceil(l / 4), ..., 1
    if (lbvh_delta(keys, n, i, i + (s + t) * d) > delta_node)
      s += t;
    if (t <= 1)
      break;
  }
  const int gamma = i + s * d + ((d < 0) ? -1 : 0);
  return {uint32_t((i < j) ? i : j), uint32_t((i < j) ? j : i), uint32_t(gamma), 0u};
}

// Boxes the collapse weighs: of a sorted position (its traversal triangle) and of a radix node (LbvhBoxes::lo / hi, bottom-up pass).
struct LbvhBoxes {
  const BvhTri* tris;
  const f3* lo;
  const f3* hi;
};

ETX_HD void lbvh_triangle_box(const BvhTri* tris, uint32_t position, f3& lo, f3& hi) {
  const BvhTri& t = tris[position];
  const f3 v0 = {t.v0_index.x, t.v0_index.y, t.v0_index.z};
  const f3 v1 = v0 + f3{t.e1_flags.x, t.e1_flags.y, t.e1_flags.z}, v2 = v0 + f3{t.e2_mat.x, t.e2_mat.y, t.e2_mat.z};
  lo = fmin3(v0, fmin3(v1, v2)), hi = fmax3(v0, fmax3(v1, v2));
}

// Box of radix node `i` from its two children (leaf children: their triangle's box); the bottom-up pass calls it once both are known.
ETX_HD void lbvh_join_children(const LbvhNode* radix, const BvhTri* tris, f3* lo, f3* hi, uint32_t i) {
  const LbvhNode nd = radix[i];
  f3 llo, lhi, rlo, rhi;
  if (nd.first == nd.split)
    lbvh_triangle_box(tris, nd.first, llo, lhi);
  else
    llo = lo[nd.split], lhi = hi[nd.split];
  if (nd.split + 1u == nd.last)
    lbvh_triangle_box(tris, nd.last, rlo, rhi);
  else
    rlo = lo[nd.split + 1u], rhi = hi[nd.split + 1u];
  lo[i] = fmin3(llo, rlo), hi[i] = fmax3(lhi, rhi);
}

// Half the surface area. The decisions below compare these numbers, and the host emulation must take the decisions the kernels
// take: no contraction into fused multiply-adds (the two compilers would round differently).
ETX_HD float lbvh_half_area(const f3& lo, const f3& hi) {
#pragma clang fp contract(off)
  const f3 d = hi - lo;
  const float xy = d.x * d.y, yz = d.y * d.z, zx = d.z * d.x;
  return (xy + yz) + zx;
}

struct LbvhKid {
  uint32_t first, last, node;  // sorted positions [first, last]; the radix node that covers them (when more than one)
};

ETX_HD float lbvh_kid_area(const LbvhBoxes& boxes, const LbvhKid& kid) {
  f3 lo, hi;
  if (kid.first == kid.last)
    lbvh_triangle_box(boxes.tris, kid.first, lo, hi);
  else
    lo = boxes.lo[kid.node], hi = boxes.hi[kid.node];
  return lbvh_half_area(lo, hi);
}

// A range becomes a leaf when it is one triangle, or at most kLbvhLeafMax and not cheaper to split once more (surface area
// heuristic with the cost of a node visit = one triangle test, host_scene.cpp Builder::split).
ETX_HD bool lbvh_is_leaf(const LbvhNode* radix, const LbvhBoxes& boxes, const LbvhKid& kid) {
#pragma clang fp contract(off)
  const uint32_t size = kid.last - kid.first + 1u;
  if (size == 1u)
    return true;
  if (size > kLbvhLeafMax)
    return false;
  const LbvhNode nd = radix[kid.node];
  const LbvhKid left = {nd.first, nd.split, nd.split}, right = {nd.split + 1u, nd.last, nd.split + 1u};
  const float area = lbvh_kid_area(boxes, kid);
  const float leaf_cost = area * float(size);
  const float left_cost = lbvh_kid_area(boxes, left) * float(left.last - left.first + 1u);
  const float right_cost = lbvh_kid_area(boxes, right) * float(right.last - right.first + 1u);
  const float split_cost = (left_cost + right_cost) + area;
  return leaf_cost <= split_cost;
}

// One four-wide node from inner node `source` of the radix tree: a leaf child is final (~((first << 3) | (count - 1))), a child
// that stays an inner node is returned as its radix node index in `inner[k]` (kInvalid otherwise) and the caller assigns the
// breadth-first index.
ETX_HD uint32_t lbvh_collapse(const LbvhNode* radix, const LbvhBoxes& boxes, uint32_t source, int32_t child[4], uint32_t inner[4]) {
  const LbvhNode root = radix[source];
  LbvhKid kids[4] = {{root.first, root.split, root.split}, {root.split + 1u, root.last, root.split + 1u}, {0u, 0u, 0u}, {0u, 0u, 0u}};
  bool leaf[4] = {lbvh_is_leaf(radix, boxes, kids[0]), lbvh_is_leaf(radix, boxes, kids[1]), true, true};
  uint32_t count = 2u;
  while (count < 4u) {
    int best = -1;
    float best_area = -1.0f;
    for (uint32_t k = 0; k < count; ++k) {
      if (leaf[k])
        continue;
      const float area = lbvh_kid_area(boxes, kids[k]);
      if (area > best_area)
        best_area = area, best = int(k);
    }
    if (best < 0)
      break;
    const LbvhNode expanded = radix[kids[best].node];
    kids[best] = {expanded.first, expanded.split, expanded.split};
    kids[count] = {expanded.split + 1u, expanded.last, expanded.split + 1u};
    leaf[best] = lbvh_is_leaf(radix, boxes, kids[best]);
    leaf[count] = lbvh_is_leaf(radix, boxes, kids[count]);
    count += 1u;
  }
  for (uint32_t k = 0; k < 4u; ++k) {
    child[k] = kBvhEmptyChild;
    inner[k] = kInvalid;
    if (k >= count)
      continue;
    if (leaf[k])
      child[k] = ~int32_t((kids[k].first << 3u) | (kids[k].last - kids[k].first));
    else
      inner[k] = kids[k].node;
  }
  return count;
}

// Traversal triangle of slot `slot` (BvhTri: v0, e1, e2, filter flags, material) from the scene tables; the triangle the slot
// holds (v0_index.w) is kept. Same arithmetic as the host builder (host_scene.cpp build_bvh).
ETX_HD void bvh_triangle_update(const DScene& scene, BvhTri* tris, uint32_t slot) {
  const uint32_t ti = lbvh_float_bits(tris[slot].v0_index.w);
  const etx_abi_triangle& t = scene.triangles[ti];
  const etx_abi_vertex &a = scene.vertices[t.i[0]], &b = scene.vertices[t.i[1]], &c = scene.vertices[t.i[2]];
  const f3 p0 = {a.pos.x, a.pos.y, a.pos.z}, p1 = {b.pos.x, b.pos.y, b.pos.z}, p2 = {c.pos.x, c.pos.y, c.pos.z};
  uint32_t flags = 0u;
  if (t.material_index < scene.material_count) {  // the filters of Raytracing::trace / trace_transmittance (rt.cxx:441-444, 503)
    const etx_abi_material& m = scene.materials[t.material_index];
    if (m.cls == ETX_MAT_VOID)
      flags |= kTriVoid;
    if (m.cls == ETX_MAT_BOUNDARY)
      flags |= kTriBoundary;
    const bool alpha_image = (m.scattering.image_index != kInvalid) && (m.scattering.image_index < scene.image_count) && ((scene.images[m.scattering.image_index].options & ETX_IMAGE_HAS_ALPHA) != 0u);
    if ((m.opacity < 1.0f) || alpha_image)
      flags |= kTriAlphaTested;
  }
  const f3 e1 = p1 - p0, e2 = p2 - p0;
  tris[slot].v0_index = make_float4(p0.x, p0.y, p0.z, lbvh_bits_float(ti));
  tris[slot].e1_flags = make_float4(e1.x, e1.y, e1.z, lbvh_bits_float(flags));
  tris[slot].e2_mat = make_float4(e2.x, e2.y, e2.z, lbvh_bits_float(t.material_index));
}

// The four child boxes of node `index` from its leaves' triangles (the vertices themselves, as the host builder bounds them) or
// from the boxes of the child node (a deeper level: already done), and the traversal stack the subtree can need (pad[0]):
// descending into one child leaves at most the other children of the node on the stack.
ETX_HD void bvh_refit_node(const DScene& scene, Bvh4Node* nodes, uint32_t index) {
  Bvh4Node& node = nodes[index];
  float lo[3][4], hi[3][4];
  uint32_t kids = 0u, deepest = 0u;
  for (uint32_t k = 0; k < 4u; ++k) {
    f3 bmin = mk3(kMaxFloat), bmax = mk3(-kMaxFloat);  // an unused slot keeps the empty box
    const int32_t child = node.child[k];
    if (child == kBvhEmptyChild) {
    } else if (child < 0) {
      kids += 1u;
      const uint32_t leaf = uint32_t(~child), leaf_first = leaf >> 3u, leaf_count = (leaf & 7u) + 1u;
      for (uint32_t s = 0; s < leaf_count; ++s) {
        const etx_abi_triangle& t = scene.triangles[lbvh_float_bits(scene.bvh_tris[leaf_first + s].v0_index.w)];
        for (uint32_t c = 0; c < 3u; ++c) {
          const etx_abi_vertex& v = scene.vertices[t.i[c]];
          const f3 p = {v.pos.x, v.pos.y, v.pos.z};
          bmin = fmin3(bmin, p), bmax = fmax3(bmax, p);
        }
      }
    } else {
      kids += 1u;
      const Bvh4Node& below = nodes[child];
      bmin = {fminf(fminf(below.lo_x.x, below.lo_x.y), fminf(below.lo_x.z, below.lo_x.w)), fminf(fminf(below.lo_y.x, below.lo_y.y), fminf(below.lo_y.z, below.lo_y.w)),
        fminf(fminf(below.lo_z.x, below.lo_z.y), fminf(below.lo_z.z, below.lo_z.w))};
      bmax = {fmaxf(fmaxf(below.hi_x.x, below.hi_x.y), fmaxf(below.hi_x.z, below.hi_x.w)), fmaxf(fmaxf(below.hi_y.x, below.hi_y.y), fmaxf(below.hi_y.z, below.hi_y.w)),
        fmaxf(fmaxf(below.hi_z.x, below.hi_z.y), fmaxf(below.hi_z.z, below.hi_z.w))};
      deepest = (below.pad[0] > deepest) ? below.pad[0] : deepest;
    }
    lo[0][k] = bmin.x, lo[1][k] = bmin.y, lo[2][k] = bmin.z;
    hi[0][k] = bmax.x, hi[1][k] = bmax.y, hi[2][k] = bmax.z;
  }
  node.lo_x = make_float4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
  node.lo_y = make_float4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]);
  node.lo_z = make_float4(lo[2][0], lo[2][1], lo[2][2], lo[2][3]);
  node.hi_x = make_float4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]);
  node.hi_y = make_float4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]);
  node.hi_z = make_float4(hi[2][0], hi[2][1], hi[2][2], hi[2][3]);
  node.pad[0] = (kids ? kids - 1u : 0u) + deepest;
}

}  // namespace etxd
