// kernels_bvh_build.h - launch wrappers of the device-side scene build (definitions in kernels_bvh_build.hip).
// Kept out of kernels.h: only the scene upload code calls them.
#pragma once

#include <hip/hip_runtime.h>
#include "dev_scene.h"

namespace etxd {

// Traversal triangles (BvhTri: v0, e1, e2, filter flags, material) re-derived from the scene's vertices, triangles and materials;
// the slot -> triangle assignment (v0_index.w) is kept. After vertex positions or material classes changed.
void launch_bvh_triangles_update(hipStream_t stream, const DScene& scene, BvhTri* tris, uint32_t count);

// One breadth-first level [first, first + count) of the BVH4: every node's four child boxes recomputed from the leaves' triangles
// (original vertices) or from the child node's boxes. Levels are refit from the deepest to the root.
void launch_bvh_refit_level(hipStream_t stream, const DScene& scene, Bvh4Node* nodes, uint32_t first, uint32_t count);

}  // namespace etxd
