// host_scene.h - host side of etx_hip_upload_scene: validates the borrowed etx::Scene, builds the BVH and the
// device tables. Everything allocated here is owned by DeviceScene and released in its destructor.
#pragma once

#include <hip/hip_runtime.h>
#include <string>
#include <vector>

#include "dev_scene.h"
#include "host_transfer.h"

namespace etxh {

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};

struct DeviceScene {
  uint32_t flat_prims = 0;          // primitives of the flat sweep (parallelograms count once), 0 = BVH traversal
  etxd::DScene host_copy = {};      // the struct as uploaded (device pointers inside)
  etxd::DScene* device = nullptr;   // device copy of host_copy
  // Three groups of device allocations: etx_hip_update_scene (host_api.cpp) rebuilds the tables and keeps the other two.
  std::vector<void*> allocations;           // material / spectrum / emitter / medium tables, the scene header
  std::vector<void*> geometry_allocations;  // vertices, triangles, BVH nodes and triangles, flat-sweep primitives
  std::vector<void*> image_allocations;     // image pixels and their sampling tables, density grids of heterogeneous media
  int alloc_group = 0;                      // where upload() files the next allocation (0 tables, 1 geometry, 2 images)
  std::vector<etxd::DImage> image_table;    // host copy of DScene::images (device pointers inside), reused by an update
  std::vector<const float*> density_grids;  // per medium: its uploaded density grid (nullptr: homogeneous)
  std::vector<uint32_t> bvh_levels;         // first node of every breadth-first level of the BVH4, then the node count (device refit)
  bool device_bvh_build = false;            // etx_hip_set_bvh_builder: the tree is built on the device (dev_lbvh.h) instead of the host's binned SAH
  double bvh_build_ms = 0.0;                // time of the last tree build (host: wall clock of build_bvh; device: HIP events)
  uint32_t film_w = 0, film_h = 0;
  float noise_threshold = 0.0f;     // Scene::noise_threshold (adaptive sampling of the path tracer)
  uint32_t bvh_depth = 0;
  bool simple_materials = false;   // only Diffuse / Translucent / Mirror / Boundary / Void / roughness-0 Conductor in use (dev_bsdf.h)
  bool bdpt_binning = false;       // bidirectional kernels: a mixed scene (simple_materials == false) whose items are split by BSDF class (kernels_bdpt.hip kPartSimple / kPartGeneral)
  bool group_general = false;      // a material of shading group kShadeGroupGeneral is in use (dev_scene.h)
  bool group_subsurface = false;   // ... of kShadeGroupSubsurface
  bool has_subsurface = false;     // a subsurface material is in use
  bool sss_media_complete = true;  // every subsurface material has a medium table entry for its walk, or derives one per entry point (bidirectional integrator)
  uint32_t medium_table_rows = 0;  // rows of the uploaded medium table (the scene's media + the entries derived from subsurface materials)
  bool has_subsurface_cb = false;  // ... of class Christensen-Burley (up to 24 exit points per vertex)
  bool generic_materials = false;  // a connectible (non delta) surface material other than Diffuse is in use
  bool needs_rgb_response = false; // spectral scene with RGB images behind spectra: apply_rgb needs the host's table (etx_hip_upload_rgb_response)
  size_t bvh_bytes = 0;
  HostTransfer* transfer = nullptr;  // the owning context's pinned transfer slots: every table of the host scene travels through them (host_transfer.h); set before build_device_scene
  uint32_t content_hash = 0;       // of the host tables the scene was built from (materials, emitters, media scalars, a sample of the vertices): etx_hip_checkpoint_*

  ~DeviceScene();
  void release();
  void release_tables();  // frees `allocations` only
  int sync_device_copy(std::string& error);  // after the host patched host_copy (CIE table): refresh the device-resident header
  void borrow(const DeviceScene& owner);  // non-owning view of the owner's device tables (helper lanes, host_api.cpp)
};

// Returns 0 or an ETX_HIP_ERROR_* code with `error` set. keep_geometry_and_images: `out` holds a scene whose geometry group
// (same vertex / triangle counts and indices) and image group (same images, same density grids) stay as they are on the device;
// only the tables are rebuilt from `scene` (etx_hip_update_scene).
int build_device_scene(const etx_abi_scene* scene, const etx_abi_camera* camera, DeviceScene& out, std::string& error, bool keep_geometry_and_images = false);

// After build_device_scene(keep): brings the kept traversal tables in line with `scene` on `stream` (kernels_bvh_build.hip). The
// traversal triangles are re-derived (filter flags follow the material classes); positions_moved (same counts and indices): the
// vertices and triangles of `scene` are copied over the device ones first and the BVH4 boxes refit level by level afterwards.
// rebuild: the tree is built again on the device from the moved vertices (linear BVH) instead of being refit.
int update_device_geometry(const etx_abi_scene* scene, DeviceScene& out, hipStream_t stream, bool positions_moved, bool rebuild, std::string& error);



// BVH build exposed for tests of the host logic (no GPU needed)
struct HostBvh {
  std::vector<etxd::BvhNode> nodes;    // the builder's binned-SAH BVH2 (kept for the invariants check)
  std::vector<etxd::Bvh4Node> nodes4;  // what the device traverses: the BVH2 collapsed to four-wide nodes, breadth first
  std::vector<uint32_t> level_offsets; // first nodes4 index of every breadth-first level, then nodes4.size()
  int32_t root4 = 0;
  uint32_t depth4 = 0;                 // levels of inner BVH4 nodes
  uint32_t stack_need = 0;             // entries the near-child-first traversal can have on its stack (exact bound over the tree)
  std::vector<etxd::BvhTri> tris;
  int32_t root = 0;
  uint32_t depth = 0;
};
void build_bvh(const etx_abi_scene* scene, HostBvh& out, bool keep_bvh2 = true);

// The device build (dev_lbvh.h) run on the host, element by element in the order the kernels' indices run: the tree the device
// WILL build, for tests without a GPU (invariants, stack bound, rays against the SAH tree).
void build_lbvh_host(const etx_abi_scene* scene, HostBvh& out);

}  // namespace etxh
