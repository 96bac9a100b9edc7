// kernels_bvh_build.hip - device side of the scene build: what Raytracing::commit_changes (sources/etx/rt/rt.cxx:58-88)
// hands to Embree - rtcSetSharedGeometryBuffer + rtcCommitScene rebuild everything on every scene change (app.cxx:368-399) -
// done on the device tables. The per-element steps are in dev_lbvh.h (shared with the host emulation the CPU tests check).
//   k_bvh_triangles_update : BvhTri slots from the scene tables (one thread per slot, 3 gathered vertices in, 48 B out)
//   k_bvh_refit_level      : BVH4 child boxes + stack bound of one breadth-first level, bottom up (one thread per node)
//   k_lbvh_keys / hipcub radix sort / k_lbvh_radix_nodes / k_lbvh_radix_boxes / k_lbvh_collapse_level : the linear BVH build
// All of it is HBM-latency bound and tiny next to an iteration: the build of a 1.2 M-triangle tree moves a few hundred MB.
#include "kernels_bvh_build.h"
#include "dev_lbvh.h"
#include "dev_bvh.h"
#include "../../include/etx_hip.h"

#include <hipcub/hipcub.hpp>

namespace etxd {

namespace {
constexpr uint32_t kBuildBlock = 256;

uint32_t blocks_for(uint32_t count) {
  return (count + kBuildBlock - 1u) / kBuildBlock;
}
}  // namespace

__global__ __launch_bounds__(kBuildBlock) void k_bvh_triangles_update(DScene scene, BvhTri* tris, uint32_t count) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < count)
    bvh_triangle_update(scene, tris, slot);
}

// one thread per triangle: three gathered vertices in, thirteen 16-byte rows out (dev_scene.h: the layout and why)
__global__ __launch_bounds__(kBuildBlock) void k_build_tri_shade(DScene scene, float4* rows, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  const etx_abi_triangle& t = scene.triangles[i];
  float4* r = rows + size_t(i) * kTriShadeStride;
  for (uint32_t k = 0; k < 3u; ++k) {
    const etx_abi_vertex& v = scene.vertices[t.i[k]];
    r[k] = make_float4(v.pos.x, v.pos.y, v.pos.z, v.tex.x);
    r[3u + k] = make_float4(v.nrm.x, v.nrm.y, v.nrm.z, v.tex.y);
    r[7u + k] = make_float4(v.tan.x, v.tan.y, v.tan.z, 0.0f);
    r[10u + k] = make_float4(v.btn.x, v.btn.y, v.btn.z, 0.0f);
  }
  r[6] = make_float4(t.geo_n.x, t.geo_n.y, t.geo_n.z, __uint_as_float(t.material_index));
}

__global__ __launch_bounds__(kBuildBlock) void k_bvh_refit_level(DScene scene, Bvh4Node* nodes, uint32_t first, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count)
    bvh_refit_node(scene, nodes, first + i);
}

__global__ __launch_bounds__(kBuildBlock) void k_lbvh_keys(DScene scene, f3 cube_min, float inv_extent, uint64_t* keys, uint32_t* values, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  keys[i] = lbvh_morton_key(scene, i, cube_min, inv_extent);
  values[i] = i;
}

// sorted triangle order -> the slots of the traversal triangles (filled in by k_bvh_triangles_update afterwards)
__global__ __launch_bounds__(kBuildBlock) void k_lbvh_assign_slots(const uint32_t* sorted_values, BvhTri* tris, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count)
    tris[i].v0_index.w = __uint_as_float(sorted_values[i]);
}

// radix node i, and who its children's parent is: `parent` of an inner child, `leaf_parent` of a child that is one sorted position
__global__ __launch_bounds__(kBuildBlock) void k_lbvh_radix_nodes(const uint64_t* sorted_keys, LbvhNode* radix, uint32_t* parent, uint32_t* leaf_parent, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1u >= count)
    return;
  const LbvhNode nd = lbvh_node(sorted_keys, int(count), int(i));
  radix[i] = nd;
  if (nd.first == nd.split)
    leaf_parent[nd.first] = i;
  else
    parent[nd.split] = i;
  if (nd.split + 1u == nd.last)
    leaf_parent[nd.last] = i;
  else
    parent[nd.split + 1u] = i;
}

// Boxes of the radix nodes, bottom-up: one thread per sorted position climbs from its parent; at every node the first thread to
// arrive stops, the second - both children are complete then - joins their boxes and goes on (Karras 2012, section 3). Nobody waits.
__global__ __launch_bounds__(kBuildBlock) void k_lbvh_radix_boxes(const LbvhNode* radix, const BvhTri* tris, const uint32_t* parent, const uint32_t* leaf_parent, uint32_t* arrivals, f3* lo, f3* hi,
  uint32_t count) {
  const uint32_t position = blockIdx.x * blockDim.x + threadIdx.x;
  if (position >= count)
    return;
  uint32_t node = leaf_parent[position];
  for (;;) {
    __threadfence();  // this thread's box of the child below is visible before its arrival is
    if (atomicAdd(arrivals + node, 1u) == 0u)
      return;
    __threadfence();
    lbvh_join_children(radix, tris, lo, hi, node);
    if (node == 0u)
      return;
    node = parent[node];
  }
}

// One breadth-first level: `queue` holds the radix nodes that become the BVH4 nodes [base, base + count); their inner children are
// appended to `next_queue` (slot from an atomic counter: the order inside a level is arbitrary) and numbered base + count + slot.
__global__ __launch_bounds__(kBuildBlock) void k_lbvh_collapse_level(const LbvhNode* radix, LbvhBoxes boxes, const uint32_t* queue, uint32_t base, uint32_t count, Bvh4Node* nodes, uint32_t* next_queue,
  uint32_t* next_count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  int32_t child[4];
  uint32_t inner[4];
  lbvh_collapse(radix, boxes, queue[i], child, inner);
  for (uint32_t k = 0; k < 4u; ++k) {
    if (inner[k] == kInvalid)
      continue;
    const uint32_t slot = atomicAdd(next_count, 1u);
    next_queue[slot] = inner[k];
    child[k] = int32_t(base + count + slot);
  }
  Bvh4Node& node = nodes[base + i];
  node.child[0] = child[0], node.child[1] = child[1], node.child[2] = child[2], node.child[3] = child[3];
  node.pad[0] = node.pad[1] = node.pad[2] = node.pad[3] = 0u;
}

template <bool kShort>  // dev_bvh.h LaneStack (32 entries in LDS) or ShortLaneStack (16)
__global__ __launch_bounds__(kBuildBlock) void k_stack_selftest(DScene scene, uint32_t depth, uint32_t* errors) {
  __shared__ int32_t s_stack[(kShort ? kShortStackDepth : kStackDepth) * kBuildBlock];
  typedef typename std::conditional<kShort, ShortLaneStack, LaneStack>::type Stack;
  Stack stack;
  if constexpr (kShort)
    stack = short_lane_stack(scene, s_stack + threadIdx.x, kBuildBlock);
  else
    stack = lane_stack(scene, s_stack + threadIdx.x, kBuildBlock);
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t sp = 0u, bad = 0u;
  for (uint32_t i = 0; i < depth; ++i)
    stack.push(sp, int32_t(lane * 131u + i * 7u + 1u));
  for (uint32_t i = depth; i-- > depth / 2u;)
    bad += (stack.pop(sp) != int32_t(lane * 131u + i * 7u + 1u)) ? 1u : 0u;
  for (uint32_t i = depth / 2u; i < depth; ++i)
    stack.push(sp, int32_t(lane * 977u + i));
  for (uint32_t i = depth; i-- > depth / 2u;)
    bad += (stack.pop(sp) != int32_t(lane * 977u + i)) ? 1u : 0u;
  for (uint32_t i = depth / 2u; i-- > 0u;)
    bad += (stack.pop(sp) != int32_t(lane * 131u + i * 7u + 1u)) ? 1u : 0u;
  if ((bad != 0u) || (sp != 0u))
    atomicAdd(errors, bad + ((sp != 0u) ? 1u : 0u));
}

void launch_stack_selftest(hipStream_t stream, int32_t* spill, uint32_t spill_lanes, uint32_t blocks, uint32_t depth, uint32_t* errors) {
  DScene scene = {};
  scene.stack_spill = spill, scene.stack_spill_lanes = spill_lanes;
  hipLaunchKernelGGL(k_stack_selftest<false>, dim3(blocks), dim3(kBuildBlock), 0, stream, scene, depth, errors);
  hipLaunchKernelGGL(k_stack_selftest<true>, dim3(blocks), dim3(kBuildBlock), 0, stream, scene, depth, errors);
}

void launch_build_tri_shade(hipStream_t stream, const DScene& scene, float4* rows, uint32_t triangle_count) {
  if (triangle_count > 0u)
    hipLaunchKernelGGL(k_build_tri_shade, dim3(blocks_for(triangle_count)), dim3(kBuildBlock), 0, stream, scene, rows, triangle_count);
}

void launch_bvh_triangles_update(hipStream_t stream, const DScene& scene, BvhTri* tris, uint32_t count) {
  if (count > 0u)
    hipLaunchKernelGGL(k_bvh_triangles_update, dim3(blocks_for(count)), dim3(kBuildBlock), 0, stream, scene, tris, count);
}

void launch_bvh_refit_level(hipStream_t stream, const DScene& scene, Bvh4Node* nodes, uint32_t first, uint32_t count) {
  if (count > 0u)
    hipLaunchKernelGGL(k_bvh_refit_level, dim3(blocks_for(count)), dim3(kBuildBlock), 0, stream, scene, nodes, first, count);
}

int lbvh_build_device(hipStream_t stream, DScene scene, f3 cube_min, float cube_extent, Bvh4Node* nodes, BvhTri* tris, LbvhResult& result, std::string& error) {
  const uint32_t n = scene.triangle_count;
  result = {};
  if ((n <= kLbvhLeafMax) || (n >= (1u << 28u)) || (!(cube_extent > 0.0f))) {
    error = "device BVH build: " + std::to_string(n) + " triangles in a cube of extent " + std::to_string(cube_extent) + " (needs more than one leaf, fewer than 2^28 triangles, a finite scene)";
    return ETX_HIP_ERROR_UNSUPPORTED;
  }
  uint64_t *keys = nullptr, *sorted_keys = nullptr;
  uint32_t *values = nullptr, *sorted_values = nullptr, *queue_a = nullptr, *queue_b = nullptr, *counter = nullptr, *parent = nullptr, *leaf_parent = nullptr, *arrivals = nullptr;
  f3 *box_lo = nullptr, *box_hi = nullptr;
  LbvhNode* radix = nullptr;
  void* sort_storage = nullptr;
  size_t sort_bytes = 0;
  hipEvent_t begin = nullptr, end = nullptr;
  // what the host reads back during the build (one counter per level, the root node at the end) arrives in PINNED memory of the build's own:
  // the library never hands HIP a pageable host pointer (host_transfer.h)
  unsigned char* pinned = nullptr;
  auto cleanup = [&]() {
    if (pinned != nullptr)
      (void)hipHostFree(pinned);
    for (void* p : {static_cast<void*>(keys), static_cast<void*>(sorted_keys), static_cast<void*>(values), static_cast<void*>(sorted_values), static_cast<void*>(queue_a),
           static_cast<void*>(queue_b), static_cast<void*>(counter), static_cast<void*>(radix), static_cast<void*>(parent), static_cast<void*>(leaf_parent), static_cast<void*>(arrivals),
           static_cast<void*>(box_lo), static_cast<void*>(box_hi), sort_storage})
      if (p != nullptr)
        (void)hipFree(p);
    if (begin != nullptr)
      (void)hipEventDestroy(begin);
    if (end != nullptr)
      (void)hipEventDestroy(end);
  };
  auto fail = [&](const char* what) {
    error = std::string("device BVH build: ") + what + " failed (" + hipGetErrorString(hipGetLastError()) + ")";
    cleanup();
    return ETX_HIP_ERROR_HIP;
  };
  if ((hipMalloc(&keys, n * sizeof(uint64_t)) != hipSuccess) || (hipMalloc(&sorted_keys, n * sizeof(uint64_t)) != hipSuccess) || (hipMalloc(&values, n * sizeof(uint32_t)) != hipSuccess) ||
      (hipMalloc(&sorted_values, n * sizeof(uint32_t)) != hipSuccess) || (hipMalloc(&queue_a, n * sizeof(uint32_t)) != hipSuccess) || (hipMalloc(&queue_b, n * sizeof(uint32_t)) != hipSuccess) ||
      (hipMalloc(&counter, sizeof(uint32_t)) != hipSuccess) || (hipMalloc(&radix, n * sizeof(LbvhNode)) != hipSuccess) || (hipMalloc(&parent, n * sizeof(uint32_t)) != hipSuccess) ||
      (hipMalloc(&leaf_parent, n * sizeof(uint32_t)) != hipSuccess) || (hipMalloc(&arrivals, n * sizeof(uint32_t)) != hipSuccess) || (hipMalloc(&box_lo, n * sizeof(f3)) != hipSuccess) ||
      (hipMalloc(&box_hi, n * sizeof(f3)) != hipSuccess))
    return fail("hipMalloc of the temporaries");
  if (hipHostMalloc(reinterpret_cast<void**>(&pinned), 256, hipHostMallocDefault) != hipSuccess)
    return fail("hipHostMalloc of the read-back words");
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys, sorted_keys, values, sorted_values, int(n), 0, 63, stream) != hipSuccess)
    return fail("sizing the radix sort");
  if (hipMalloc(&sort_storage, std::max<size_t>(sort_bytes, 16)) != hipSuccess)
    return fail("hipMalloc of the sort storage");
  if ((hipEventCreate(&begin) != hipSuccess) || (hipEventCreate(&end) != hipSuccess))
    return fail("hipEventCreate");
  (void)hipEventRecord(begin, stream);

  hipLaunchKernelGGL(k_lbvh_keys, dim3(blocks_for(n)), dim3(kBuildBlock), 0, stream, scene, cube_min, 1.0f / cube_extent, keys, values, n);
  if (hipcub::DeviceRadixSort::SortPairs(sort_storage, sort_bytes, keys, sorted_keys, values, sorted_values, int(n), 0, 63, stream) != hipSuccess)
    return fail("the radix sort");
  hipLaunchKernelGGL(k_lbvh_assign_slots, dim3(blocks_for(n)), dim3(kBuildBlock), 0, stream, sorted_values, tris, n);
  scene.bvh_tris = tris;  // the refit reads the slots' triangle indices through the scene
  scene.bvh_tri_count = n;
  launch_bvh_triangles_update(stream, scene, tris, n);
  hipLaunchKernelGGL(k_lbvh_radix_nodes, dim3(blocks_for(n)), dim3(kBuildBlock), 0, stream, sorted_keys, radix, parent, leaf_parent, n);
  if (hipMemsetAsync(arrivals, 0, n * sizeof(uint32_t), stream) != hipSuccess)
    return fail("hipMemset of the arrival counters");
  hipLaunchKernelGGL(k_lbvh_radix_boxes, dim3(blocks_for(n)), dim3(kBuildBlock), 0, stream, radix, tris, parent, leaf_parent, arrivals, box_lo, box_hi, n);
  const LbvhBoxes boxes = {tris, box_lo, box_hi};

  // collapse, level by level; the host reads one counter per level (a build step, not the render loop)
  if (hipMemsetAsync(queue_a, 0, sizeof(uint32_t), stream) != hipSuccess)  // the first queue entry: radix node 0 covers every key
    return fail("hipMemset of the root");
  uint32_t base = 0u, count = 1u;
  uint32_t *queue = queue_a, *next_queue = queue_b;
  while (count > 0u) {
    if (uint64_t(base) + count > uint64_t(n)) {
      error = "device BVH build: more nodes than triangles (internal error)";
      cleanup();
      return ETX_HIP_ERROR_HIP;
    }
    result.level_offsets.push_back(base);
    if (hipMemsetAsync(counter, 0, sizeof(uint32_t), stream) != hipSuccess)
      return fail("hipMemset of the level counter");
    hipLaunchKernelGGL(k_lbvh_collapse_level, dim3(blocks_for(count)), dim3(kBuildBlock), 0, stream, radix, boxes, queue, base, count, nodes, next_queue, counter);
    if ((hipMemcpyAsync(pinned, counter, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) || (hipStreamSynchronize(stream) != hipSuccess))
      return fail("reading the level counter");
    uint32_t next = 0u;
    memcpy(&next, pinned, sizeof(uint32_t));
    base += count;
    count = next;
    std::swap(queue, next_queue);
  }
  result.node_count = base;
  result.level_offsets.push_back(base);
  result.depth = uint32_t(result.level_offsets.size()) - 1u;
  result.root = 0;
  for (size_t level = result.level_offsets.size(); level-- > 1u;)
    launch_bvh_refit_level(stream, scene, nodes, result.level_offsets[level - 1u], result.level_offsets[level] - result.level_offsets[level - 1u]);
  static_assert(sizeof(Bvh4Node) <= 256, "the pinned read-back area holds one node");
  Bvh4Node root_node;
  (void)hipEventRecord(end, stream);
  if ((hipMemcpyAsync(pinned, nodes, sizeof(Bvh4Node), hipMemcpyDeviceToHost, stream) != hipSuccess) || (hipStreamSynchronize(stream) != hipSuccess) || (hipGetLastError() != hipSuccess))
    return fail("the box pass");
  memcpy(&root_node, pinned, sizeof(Bvh4Node));
  result.stack_need = root_node.pad[0];
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, begin, end);
  result.milliseconds = ms;
  cleanup();
  return 0;
}

}  // namespace etxd
