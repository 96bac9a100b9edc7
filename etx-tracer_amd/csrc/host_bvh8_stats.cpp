// host_bvh8_stats.cpp - host-only test entry point: the ENCODED eight-wide tree walked through the node function the kernels call.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/etx_hip.h"
#include "host_scene.h"

using etxd::BvhTri;
using etxd::kBvhEmptyChild;

// ---------------------------------------------------------------------------------------------------------------
// The encoded eight-wide tree (host_scene.cpp encode_bvh8 -> dev_scene.h Bvh8Node) walked on the host through dev_bvh8.h bvh8_visit - the
// function the kernels call, byte decoding and folded slab test included. Same outputs as etx_hip_host_bvh_study.
#include "dev_bvh8.h"

namespace {
struct HostStack {
  std::vector<int32_t>* data;
  void push(uint32_t& sp, int32_t v) const {
    if (sp == data->size())
      data->resize(data->size() * 2u);
    (*data)[sp++] = v;
  }
  int32_t pop(uint32_t& sp) const {
    return (*data)[--sp];
  }
};
}  // namespace

extern "C" int etx_hip_host_bvh8_stats(const etx_abi_scene* scene, int occlusion, const float* rays_8f, uint64_t count, uint64_t out[8], float* hits_2f) {
  if ((scene == nullptr) || (rays_8f == nullptr) || (out == nullptr))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  etxh::HostBvh bvh;
  etxh::build_bvh(scene, bvh, true);
  for (int i = 0; i < 8; ++i)
    out[i] = 0;
  if (bvh.tris.empty())
    return ETX_HIP_OK;
  etxh::HostBvh8 wide;
  std::string error;
  if (etxh::encode_bvh8(bvh, wide, error) == false)
    return ETX_HIP_ERROR_STATE;
  out[4] = wide.nodes.size(), out[5] = uint64_t(wide.levels) | (uint64_t(wide.stack_need) << 16u);  // levels | exact stack bound << 16
  std::vector<int32_t> storage(512);
  const HostStack stack = {&storage};
  for (uint64_t r = 0; r < count; ++r) {
    const float* q = rays_8f + 8 * r;
    const etxd::f3 o = {q[0], q[1], q[2]}, d = {q[4], q[5], q[6]};
    const etxd::f3 inv = {etxd::bvh8_reciprocal(d.x), etxd::bvh8_reciprocal(d.y), etxd::bvh8_reciprocal(d.z)};
    const float tmin = q[3];
    float best = q[7];
    bool hit = false;
    float hit_triangle = 0.0f;
    uint32_t sp = 0;
    uint64_t visits = 0, visits_at_last_improvement = 0;
    int32_t cur = wide.root;
    while (cur != kBvhEmptyChild) {
      if (cur >= 0) {
        visits++;
        const etxd::Bvh8Words words = etxd::bvh8_load(reinterpret_cast<const uint4*>(&wide.nodes[size_t(cur)]));
        const int32_t next = occlusion ? etxd::bvh8_visit<false>(words, o, inv, tmin, best, stack, sp) : etxd::bvh8_visit<true>(words, o, inv, tmin, best, stack, sp);
        out[3] = std::max<uint64_t>(out[3], sp);
        cur = (next != kBvhEmptyChild) ? next : (sp ? stack.pop(sp) : kBvhEmptyChild);
      } else {
        const uint32_t leaf = uint32_t(~cur), first = leaf >> 3, n = (leaf & 7u) + 1u;
        bool stop = false;
        for (uint32_t i = first; i < first + n; ++i) {
          out[1]++;
          const BvhTri& tr = bvh.tris[i];
          const float e1[3] = {tr.e1_flags.x, tr.e1_flags.y, tr.e1_flags.z}, e2[3] = {tr.e2_mat.x, tr.e2_mat.y, tr.e2_mat.z};
          const float p[3] = {d.y * e2[2] - d.z * e2[1], d.z * e2[0] - d.x * e2[2], d.x * e2[1] - d.y * e2[0]};
          const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
          if (det == 0.0f)
            continue;
          const float s[3] = {o.x - tr.v0_index.x, o.y - tr.v0_index.y, o.z - tr.v0_index.z};
          const float u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) / det;
          const float qv[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
          const float v = (d.x * qv[0] + d.y * qv[1] + d.z * qv[2]) / det, tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
          if ((u >= 0.0f) && (v >= 0.0f) && (u + v <= 1.0f) && (tt >= tmin) && (tt <= best)) {
            hit = true, hit_triangle = tr.v0_index.w;
            visits_at_last_improvement = visits;
            if (occlusion) {
              stop = true;  // any hit ends an occlusion query; `best` stays the segment's end
              break;
            }
            best = tt;
          }
        }
        cur = (stop || (sp == 0u)) ? kBvhEmptyChild : stack.pop(sp);
      }
    }
    out[0] += visits;
    out[6] += visits_at_last_improvement;
    out[7] = std::max<uint64_t>(out[7], visits);
    out[2] += hit ? 1u : 0u;
    if (hits_2f != nullptr) {
      const uint32_t miss = 0xffffffffu;
      hits_2f[2 * r + 0] = (hit && (occlusion == 0)) ? best : 0.0f;
      if (hit)
        hits_2f[2 * r + 1] = hit_triangle;
      else
        memcpy(hits_2f + 2 * r + 1, &miss, 4);
    }
  }
  return ETX_HIP_OK;
}
