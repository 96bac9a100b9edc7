// bvh_study.cpp - a TOOL, not part of the product library (built on demand by tools/bvh_study.py into tools/bvh_study_src/libetx_bvh_study.so,
// linked against libetx_hip.so for the host tree builder). Host-only: node formats the traversal kernels could read instead of the 128-byte float BVH4 node, built
// from the same binned-SAH BVH2 and walked on the host exactly as a kernel would walk them, to count what a ray costs in each:
//   width 4 or 8   the BVH2 collapsed to that many children per node (largest child first, like build_bvh's BVH4 collapse)
//   quantised      child boxes as 8-bit offsets in the node's own frame (origin + power-of-two scale per axis), rounded outwards:
//                  the decoded box contains the exact one, so a traversal visits a superset of nodes and finds the same closest hit
//   order          children visited nearest first with the rest pushed sorted (what dev_bvh.h bvh_closest does for four children)
//                  or nearest first with the rest pushed as they come (what an eight-wide kernel can afford)
// Reported per ray set: node visits, triangle tests, the deepest stack, the longest chain of dependent node fetches a ray makes and
// its sum, and the hits (tests/test_host_bvh_study.py: the same hits for every format). No device code reads these formats yet;
// DESIGN.md 7 quotes the numbers as the basis for the next traversal kernel.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/etx_hip.h"
#include "../../etx-tracer_amd/csrc/host_scene.h"

namespace {

using etxd::BvhNode;
using etxd::BvhTri;
using etxd::kBvhEmptyChild;
constexpr float kMaxF = 3.402823466e+38f;

struct Box {
  float lo[3], hi[3];
};

struct WideNode {
  uint32_t count = 0;
  int32_t child[8];
  Box box[8];  // what the traversal tests: exact, or decoded from the quantised form
};

float half_area(const Box& b) {
  const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
  return x * y + y * z + z * x;
}

void child_boxes(const BvhNode& n, Box& b0, Box& b1) {
  b0 = {{n.lo0_hi0x.x, n.lo0_hi0x.y, n.lo0_hi0x.z}, {n.lo0_hi0x.w, n.hi0yz_lo1xy.x, n.hi0yz_lo1xy.y}};
  b1 = {{n.hi0yz_lo1xy.z, n.hi0yz_lo1xy.w, n.lo1z_hi1.x}, {n.lo1z_hi1.y, n.lo1z_hi1.z, n.lo1z_hi1.w}};
}

// Outward 8-bit quantisation of the children of one node in the node's frame. Returns false if a decoded box fails to contain its
// exact box (never expected: the rounding is outward and checked in double precision).
bool quantise(WideNode& node) {
  double lo[3] = {kMaxF, kMaxF, kMaxF}, hi[3] = {-kMaxF, -kMaxF, -kMaxF};
  for (uint32_t k = 0; k < node.count; ++k)
    for (int a = 0; a < 3; ++a)
      lo[a] = std::min<double>(lo[a], node.box[k].lo[a]), hi[a] = std::max<double>(hi[a], node.box[k].hi[a]);
  bool ok = true;
  for (int a = 0; a < 3; ++a) {
    const float origin = float(lo[a]);  // exact: one of the children's floats
    const double extent = hi[a] - lo[a];
    int e = -126;
    if (extent > 0.0) {
      e = int(std::ceil(std::log2(extent / 255.0)));
      while (std::ldexp(255.0, e) < extent)  // log2 rounding
        ++e;
      e = std::max(e, -126);
    }
    const double scale = std::ldexp(1.0, e);
    for (uint32_t k = 0; k < node.count; ++k) {
      const double ql = std::floor((double(node.box[k].lo[a]) - double(origin)) / scale);
      const double qh = std::ceil((double(node.box[k].hi[a]) - double(origin)) / scale);
      const uint32_t il = uint32_t(std::min(255.0, std::max(0.0, ql))), ih = uint32_t(std::min(255.0, std::max(0.0, qh)));
      // decode the way a kernel would: float(q) * scale + origin in single precision (fma)
      float dl = std::fma(float(il), float(scale), origin), dh = std::fma(float(ih), float(scale), origin);
      // the single-precision decode may land a hair inside: step outwards in units of the last place
      while (dl > node.box[k].lo[a])
        dl = std::nextafter(dl, -kMaxF);
      while (dh < node.box[k].hi[a])
        dh = std::nextafter(dh, kMaxF);
      ok = ok && (dl <= node.box[k].lo[a]) && (dh >= node.box[k].hi[a]) && (ih <= 255u);
      node.box[k].lo[a] = dl, node.box[k].hi[a] = dh;
    }
  }
  return ok;
}

struct WideTree {
  std::vector<WideNode> nodes;  // breadth first
  int32_t root = kBvhEmptyChild;
  uint32_t depth = 0;
};

bool build_wide(const etxh::HostBvh& bvh, uint32_t width, bool quantised, WideTree& out) {
  out = {};
  if (bvh.root < 0) {  // one leaf (or nothing)
    out.root = bvh.tris.empty() ? kBvhEmptyChild : bvh.root;
    return true;
  }
  struct Pending {
    int32_t bvh2;
    uint32_t level;
  };
  std::vector<Pending> queue;
  queue.push_back({bvh.root, 1u});
  out.root = 0;
  bool ok = true;
  for (size_t head = 0; head < queue.size(); ++head) {
    const Pending item = queue[head];
    out.depth = std::max(out.depth, item.level);
    int32_t kids[8];
    Box boxes[8];
    uint32_t kid_count = 2;
    kids[0] = bvh.nodes[item.bvh2].child0, kids[1] = bvh.nodes[item.bvh2].child1;
    child_boxes(bvh.nodes[item.bvh2], boxes[0], boxes[1]);
    while (kid_count < width) {
      int best = -1;
      float best_area = -1.0f;
      for (uint32_t k = 0; k < kid_count; ++k) {
        if (kids[k] < 0)
          continue;  // leaf
        const float area = half_area(boxes[k]);
        if (area > best_area)
          best_area = area, best = int(k);
      }
      if (best < 0)
        break;
      const BvhNode& expanded = bvh.nodes[kids[best]];
      kids[best] = expanded.child0, kids[kid_count] = expanded.child1;
      child_boxes(expanded, boxes[best], boxes[kid_count]);
      kid_count++;
    }
    WideNode node;
    node.count = kid_count;
    for (uint32_t k = 0; k < kid_count; ++k) {
      node.box[k] = boxes[k];
      if (kids[k] < 0) {
        node.child[k] = kids[k];
      } else {
        node.child[k] = int32_t(queue.size());
        queue.push_back({kids[k], item.level + 1u});
      }
    }
    if (quantised)
      ok = quantise(node) && ok;
    out.nodes.push_back(node);
  }
  return ok;
}

}  // namespace

extern "C" {

// out[0] node visits, out[1] triangle tests, out[2] rays that hit, out[3] deepest stack, out[4] nodes of the tree, out[5] inner levels,
// out[6] sum over rays of the node visits made BEFORE the visit that found the final hit's leaf was reached (the dependent chain a
// latency-bound kernel waits for is the whole visit sequence of a ray: out[0] / rays; this entry separates search from confirmation),
// out[7] the largest number of node visits of a single ray. hits_2f (optional): t and triangle index bits per ray, 0 / 0xffffffff for a miss.
int etx_bvh_study(const etx_abi_scene* scene, uint32_t width, int quantised, int sorted_pushes, const float* rays_8f, uint64_t count, uint64_t out[8], float* hits_2f) {
  if ((scene == nullptr) || (rays_8f == nullptr) || (out == nullptr) || ((width != 4u) && (width != 8u)))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  etxh::HostBvh bvh;
  etxh::build_bvh(scene, bvh, true);
  for (int i = 0; i < 8; ++i)
    out[i] = 0;
  if (bvh.tris.empty())
    return ETX_HIP_OK;
  WideTree tree;
  if (build_wide(bvh, width, quantised != 0, tree) == false)
    return ETX_HIP_ERROR_STATE;  // a decoded box did not contain its exact box
  out[4] = tree.nodes.size(), out[5] = tree.depth;
  std::vector<int32_t> stack(512);
  for (uint64_t r = 0; r < count; ++r) {
    const float* q = rays_8f + 8 * r;
    const float o[3] = {q[0], q[1], q[2]}, d[3] = {q[4], q[5], q[6]};
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float tmin = q[3];
    float best = q[7];
    bool hit = false;
    float hit_triangle = 0.0f;
    size_t sp = 0;
    uint64_t visits = 0, visits_at_last_improvement = 0;
    int32_t cur = tree.root;
    while (cur != kBvhEmptyChild) {
      if (cur >= 0) {
        visits++;
        const WideNode& nd = tree.nodes[size_t(cur)];
        float t[8];
        int32_t c[8];
        uint32_t hits = 0;
        for (uint32_t k = 0; k < nd.count; ++k) {
          float t_enter = tmin, t_exit = best;
          for (int a = 0; a < 3; ++a) {
            const float t0 = (nd.box[k].lo[a] - o[a]) * inv[a], t1 = (nd.box[k].hi[a] - o[a]) * inv[a];
            t_enter = std::max(t_enter, std::min(t0, t1));
            t_exit = std::min(t_exit, std::max(t0, t1));
          }
          if (t_enter <= t_exit * 1.0000004f)
            t[hits] = t_enter, c[hits] = nd.child[k], hits++;
        }
        if (hits == 0u) {
          cur = sp ? stack[--sp] : kBvhEmptyChild;
          continue;
        }
        if (sorted_pushes) {
          for (uint32_t i = 0; i < hits; ++i)
            for (uint32_t j = i + 1; j < hits; ++j)
              if (t[j] < t[i])
                std::swap(t[i], t[j]), std::swap(c[i], c[j]);
        } else {  // the nearest child to slot 0, the others stay in node order
          uint32_t nearest = 0;
          for (uint32_t i = 1; i < hits; ++i)
            if (t[i] < t[nearest])
              nearest = i;
          std::swap(t[0], t[nearest]), std::swap(c[0], c[nearest]);
        }
        for (uint32_t k = hits; k-- > 1u;) {
          if (sp == stack.size())
            stack.resize(stack.size() * 2);
          stack[sp++] = c[k];
        }
        out[3] = std::max<uint64_t>(out[3], sp);
        cur = c[0];
      } else {
        const uint32_t leaf = uint32_t(~cur), first = leaf >> 3, n = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + n; ++i) {
          out[1]++;
          const BvhTri& tr = bvh.tris[i];
          const float e1[3] = {tr.e1_flags.x, tr.e1_flags.y, tr.e1_flags.z}, e2[3] = {tr.e2_mat.x, tr.e2_mat.y, tr.e2_mat.z};
          const float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
          const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
          if (det == 0.0f)
            continue;
          const float s[3] = {o[0] - tr.v0_index.x, o[1] - tr.v0_index.y, o[2] - tr.v0_index.z};
          const float u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) / det;
          const float qv[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
          const float v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) / det, tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
          if ((u >= 0.0f) && (v >= 0.0f) && (u + v <= 1.0f) && (tt >= tmin) && (tt <= best)) {
            best = tt, hit = true, hit_triangle = tr.v0_index.w;
            visits_at_last_improvement = visits;
          }
        }
        cur = sp ? stack[--sp] : kBvhEmptyChild;
      }
    }
    out[0] += visits;
    out[6] += visits_at_last_improvement;
    out[7] = std::max<uint64_t>(out[7], visits);
    out[2] += hit ? 1u : 0u;
    if (hits_2f != nullptr) {
      const uint32_t miss = 0xffffffffu;
      hits_2f[2 * r + 0] = hit ? best : 0.0f;
      if (hit)
        hits_2f[2 * r + 1] = hit_triangle;
      else
        memcpy(hits_2f + 2 * r + 1, &miss, 4);
    }
  }
  return ETX_HIP_OK;
}

}  // extern "C"

