// kernels_bvh_build.h - launch wrappers of the device-side scene build (definitions in kernels_bvh_build.hip).
// Kept out of kernels.h: only the scene upload code calls them.
#pragma once

#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "dev_scene.h"

namespace etxd {

// Traversal triangles (BvhTri: v0, e1, e2, filter flags, material) re-derived from the scene's vertices, triangles and materials;
// the slot -> triangle assignment (v0_index.w) is kept. After vertex positions or material classes changed.
void launch_bvh_triangles_update(hipStream_t stream, const DScene& scene, BvhTri* tris, uint32_t count);

// The shading records (DScene::tri_shade, dev_scene.h kTriShadeStride rows per triangle) from the scene's vertex and triangle tables: after every
// upload of those tables.
void launch_build_tri_shade(hipStream_t stream, const DScene& scene, float4* rows, uint32_t triangle_count);

// One breadth-first level [first, first + count) of the BVH4: every node's four child boxes recomputed from the leaves' triangles
// (original vertices) or from the child node's boxes, and the stack bound of its subtree (pad[0]). Levels are refit from the
// deepest to the root.
void launch_bvh_refit_level(hipStream_t stream, const DScene& scene, Bvh4Node* nodes, uint32_t first, uint32_t count);

// The whole tree built on the device (dev_lbvh.h): `scene` holds device vertices / triangles / materials / images and their counts;
// `nodes` (capacity: triangle_count entries) and `tris` (triangle_count entries) are device buffers that receive the BVH4 and the
// traversal triangles. Temporaries live for the call. Returns 0 or an ETX_HIP_ERROR_* code with `error` set.
struct LbvhResult {
  uint32_t node_count = 0, depth = 0, stack_need = 0;
  int32_t root = 0;
  std::vector<uint32_t> level_offsets;  // first node of every breadth-first level, then node_count
  double milliseconds = 0.0;            // device time of the build (HIP events)
};
int lbvh_build_device(hipStream_t stream, DScene scene, f3 cube_min, float cube_extent, Bvh4Node* nodes, BvhTri* tris, LbvhResult& result, std::string& error);

// Self test of the traversal stack (dev_bvh.h LaneStack: kStackDepth entries per lane in LDS, the rest in the global spill area):
// every lane of a traversal-sized grid pushes `depth` values, pops half, pushes again and pops everything, checking each value. A
// real ray never gets near the spill (the bound is a worst case over the tree), so this is what exercises it. Returns the number of
// mismatches in *errors (device memory).
void launch_stack_selftest(hipStream_t stream, int32_t* spill, uint32_t spill_lanes, uint32_t blocks, uint32_t depth, uint32_t* errors);

}  // namespace etxd
