// kernels_bvh_build.hip - device side of the scene build: what Raytracing::commit_changes (sources/etx/rt/rt.cxx:58-88)
// hands to Embree - rtcSetSharedGeometryBuffer + rtcCommitScene rebuild everything on every scene change (app.cxx:368-399) -
// done in place on the device tables when only vertex positions or materials changed.
//   k_bvh_triangles_update : BvhTri slots from the scene tables (one thread per slot, 3 gathered vertices in, 48 B out)
//   k_bvh_refit_level      : BVH4 child boxes of one breadth-first level, bottom up (one thread per node; a leaf reads its <= 8
//                            triangles' vertices, an inner child the 96 B of boxes of its node). Both are HBM-latency bound and
//                            tiny next to an iteration: a refit of the 1.2 M-triangle tree moves ~200 MB.
#include "kernels_bvh_build.h"
#include "dev_math.h"

namespace etxd {

namespace {
constexpr uint32_t kBuildBlock = 256;

ETX_DEV f3 vertex_position(const DScene& scene, uint32_t index) {
  const etx_abi_vertex& v = scene.vertices[index];
  return {v.pos.x, v.pos.y, v.pos.z};
}
}  // namespace

__global__ __launch_bounds__(kBuildBlock) void k_bvh_triangles_update(DScene scene, BvhTri* tris, uint32_t count) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= count)
    return;
  const uint32_t ti = __float_as_uint(tris[slot].v0_index.w);
  const etx_abi_triangle& t = scene.triangles[ti];
  const f3 p0 = vertex_position(scene, t.i[0]), p1 = vertex_position(scene, t.i[1]), p2 = vertex_position(scene, t.i[2]);
  uint32_t flags = 0u;
  if (t.material_index < scene.material_count) {  // the filters of Raytracing::trace / trace_transmittance (host_scene.cpp build_bvh)
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
  tris[slot].v0_index = make_float4(p0.x, p0.y, p0.z, __uint_as_float(ti));
  tris[slot].e1_flags = make_float4(e1.x, e1.y, e1.z, __uint_as_float(flags));
  tris[slot].e2_mat = make_float4(e2.x, e2.y, e2.z, __uint_as_float(t.material_index));
}

__global__ __launch_bounds__(kBuildBlock) void k_bvh_refit_level(DScene scene, Bvh4Node* nodes, uint32_t first, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  Bvh4Node& node = nodes[first + i];
  float lo[3][4], hi[3][4];
  for (uint32_t k = 0; k < 4u; ++k) {
    f3 bmin = mk3(kMaxFloat), bmax = mk3(-kMaxFloat);  // an unused slot keeps the empty box the builder gave it
    const int32_t child = node.child[k];
    if (child == kBvhEmptyChild) {
    } else if (child < 0) {
      const uint32_t leaf = uint32_t(~child), leaf_first = leaf >> 3u, leaf_count = (leaf & 7u) + 1u;
      for (uint32_t s = 0; s < leaf_count; ++s) {
        const etx_abi_triangle& t = scene.triangles[__float_as_uint(scene.bvh_tris[leaf_first + s].v0_index.w)];
        for (uint32_t c = 0; c < 3u; ++c) {  // the vertices themselves, as the host builder bounds them (not v0 + e: one rounding off)
          const f3 p = vertex_position(scene, t.i[c]);
          bmin = fmin3(bmin, p), bmax = fmax3(bmax, p);
        }
      }
    } else {
      const Bvh4Node& below = nodes[child];  // a deeper level: already refit
      bmin = {fminf(fminf(below.lo_x.x, below.lo_x.y), fminf(below.lo_x.z, below.lo_x.w)), fminf(fminf(below.lo_y.x, below.lo_y.y), fminf(below.lo_y.z, below.lo_y.w)),
        fminf(fminf(below.lo_z.x, below.lo_z.y), fminf(below.lo_z.z, below.lo_z.w))};
      bmax = {fmaxf(fmaxf(below.hi_x.x, below.hi_x.y), fmaxf(below.hi_x.z, below.hi_x.w)), fmaxf(fmaxf(below.hi_y.x, below.hi_y.y), fmaxf(below.hi_y.z, below.hi_y.w)),
        fmaxf(fmaxf(below.hi_z.x, below.hi_z.y), fmaxf(below.hi_z.z, below.hi_z.w))};
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
}

void launch_bvh_triangles_update(hipStream_t stream, const DScene& scene, BvhTri* tris, uint32_t count) {
  if (count > 0u)
    hipLaunchKernelGGL(k_bvh_triangles_update, dim3((count + kBuildBlock - 1u) / kBuildBlock), dim3(kBuildBlock), 0, stream, scene, tris, count);
}

void launch_bvh_refit_level(hipStream_t stream, const DScene& scene, Bvh4Node* nodes, uint32_t first, uint32_t count) {
  if (count > 0u)
    hipLaunchKernelGGL(k_bvh_refit_level, dim3((count + kBuildBlock - 1u) / kBuildBlock), dim3(kBuildBlock), 0, stream, scene, nodes, first, count);
}

}  // namespace etxd
