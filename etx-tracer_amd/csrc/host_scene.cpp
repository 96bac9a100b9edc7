// host_scene.cpp - scene upload for the gfx950 backend (replaces Raytracing::commit_changes, sources/etx/rt/rt.cxx:58-88,
// and follows the author's disabled device-upload sketch rt.cxx:141-238: deep copy + pointer patching).
#include "host_scene.h"
#include "kernels_bvh_build.h"
#include "dev_lbvh.h"
#include "dev_bsdf.h"
#include "dev_bvh.h"
#include "../../include/etx_hip.h"
#include "tuning_knobs.h"

#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <future>
#include <thread>
#include <numeric>

namespace etxh {

using namespace etxd;

DeviceScene::~DeviceScene() {
  release();
}

void DeviceScene::release() {
  release_tables();
  for (void* p : geometry_allocations)
    (void)hipFree(p);
  for (void* p : image_allocations)
    (void)hipFree(p);
  geometry_allocations.clear();
  image_allocations.clear();
  image_table.clear();
  density_grids.clear();
  bvh_levels.clear();
  host_copy = {};
}

void DeviceScene::release_tables() {
  for (void* p : allocations)
    (void)hipFree(p);
  allocations.clear();
  device = nullptr;
}

int DeviceScene::sync_device_copy(std::string& error) {
  if (device == nullptr)
    return 0;
  if (transfer == nullptr) {
    error = "internal: the scene has no host transfer object";
    return ETX_HIP_ERROR_STATE;
  }
  return transfer->to_device(device, &host_copy, sizeof(DScene), nullptr, error);
}

void DeviceScene::borrow(const DeviceScene& owner) {
  release();
  flat_prims = owner.flat_prims;
  host_copy = owner.host_copy;
  device = owner.device;
  film_w = owner.film_w, film_h = owner.film_h;
  noise_threshold = owner.noise_threshold;
  bvh_depth = owner.bvh_depth;
  simple_materials = owner.simple_materials;
  bdpt_binning = owner.bdpt_binning;
  group_general = owner.group_general, group_subsurface = owner.group_subsurface;
  has_subsurface = owner.has_subsurface;
  has_subsurface_cb = owner.has_subsurface_cb;
  sss_media_complete = owner.sss_media_complete;
  medium_table_rows = owner.medium_table_rows;
  generic_materials = owner.generic_materials;
  needs_rgb_response = owner.needs_rgb_response;
  bvh_bytes = owner.bvh_bytes;
  content_hash = owner.content_hash;
}

namespace {

uint32_t build_threads() {  // ETX_HIP_BVH_BUILD_THREADS=1 keeps the host build on one thread
  const char* e = getenv("ETX_HIP_BVH_BUILD_THREADS");
  return e ? uint32_t(std::max(1, atoi(e))) : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
}

// body(begin, end) over [0, count) in contiguous chunks, one per thread (per-element loops of the host build of large scenes)
template <class Body>
void parallel_chunks(uint32_t count, uint32_t threads, Body body) {
  if ((threads <= 1u) || (count < 65536u)) {
    body(0u, count);
    return;
  }
  const uint32_t chunk = (count + threads - 1u) / threads;
  std::vector<std::thread> workers;
  for (uint32_t begin = chunk; begin < count; begin += chunk)
    workers.emplace_back(body, begin, std::min(count, begin + chunk));
  body(0u, std::min(count, chunk));
  for (std::thread& w : workers)
    w.join();
}

struct Builder {
  struct Prim {
    f3 bmin, bmax, centroid;
    uint32_t index;
  };
  struct TmpNode {
    f3 bmin, bmax;
    int32_t left = -1, right = -1;  // children (TmpNode index) or -1 for leaf
    uint32_t first = 0, count = 0;
  };
  Prim* prims = nullptr;  // the primitives of the whole scene (owner: build_bvh); a builder works on its range of them
  std::vector<TmpNode> nodes;
  uint32_t max_depth = 0;
  float traversal_cost = 1.0f;

  static float half_area(const f3& mn, const f3& mx) {
    f3 d = mx - mn;
    return d.x * d.y + d.y * d.z + d.z * d.x;
  }
  static float axis(const f3& v, int a) {
    return a == 0 ? v.x : (a == 1 ? v.y : v.z);
  }

  // Bounds of the range and, unless it becomes a leaf, its binned-SAH split: the range is partitioned in place, [first, mid) goes left.
  bool split(uint32_t first, uint32_t count, f3& mn, f3& mx, uint32_t& mid) {
    Prim* const prims = this->prims;
    mn = mk3(kMaxFloat), mx = mk3(-kMaxFloat);
    f3 cmn = mk3(kMaxFloat), cmx = mk3(-kMaxFloat);
    for (uint32_t i = first; i < first + count; ++i) {
      mn = fmin3(mn, prims[i].bmin), mx = fmax3(mx, prims[i].bmax);
      cmn = fmin3(cmn, prims[i].centroid), cmx = fmax3(cmx, prims[i].centroid);
    }
    constexpr uint32_t kMaxLeaf = 4;  // leaf encoding allows 8
    if (count <= 1)
      return false;

    constexpr int kBins = 16;
    float best_cost = kMaxFloat;
    int best_axis = -1, best_split = 0;
    for (int a = 0; a < 3; ++a) {
      float lo = axis(cmn, a), hi = axis(cmx, a);
      if (!(hi - lo > 0.0f))
        continue;
      struct Bin {
        f3 mn = mk3(kMaxFloat), mx = mk3(-kMaxFloat);
        uint32_t n = 0;
      } bins[kBins];
      float scale = float(kBins) / (hi - lo);
      for (uint32_t i = first; i < first + count; ++i) {
        int b = std::min(kBins - 1, int((axis(prims[i].centroid, a) - lo) * scale));
        bins[b].n++, bins[b].mn = fmin3(bins[b].mn, prims[i].bmin), bins[b].mx = fmax3(bins[b].mx, prims[i].bmax);
      }
      float la[kBins - 1], ra[kBins - 1];
      uint32_t ln[kBins - 1], rn[kBins - 1];
      Bin l, r;
      for (int i = 0; i < kBins - 1; ++i) {
        l.n += bins[i].n, l.mn = fmin3(l.mn, bins[i].mn), l.mx = fmax3(l.mx, bins[i].mx);
        ln[i] = l.n, la[i] = l.n ? half_area(l.mn, l.mx) : 0.0f;
        int j = kBins - 1 - i;
        r.n += bins[j].n, r.mn = fmin3(r.mn, bins[j].mn), r.mx = fmax3(r.mx, bins[j].mx);
        rn[j - 1] = r.n, ra[j - 1] = r.n ? half_area(r.mn, r.mx) : 0.0f;
      }
      for (int i = 0; i < kBins - 1; ++i) {
        if ((ln[i] == 0) || (rn[i] == 0))
          continue;
        float cost = la[i] * float(ln[i]) + ra[i] * float(rn[i]);
        if (cost < best_cost)
          best_cost = cost, best_axis = a, best_split = i;
      }
    }
    mid = first + count / 2;
    if (best_axis >= 0) {
      // SAH termination: a node visit of the device traversal (fetch 128 B, four slab tests, sort, stack traffic) costs
      // about as much as `traversal_cost` triangle tests; measured on the gems scene
      float leaf_cost = half_area(mn, mx) * float(count);
      if ((count <= kMaxLeaf) && (best_cost + traversal_cost * half_area(mn, mx) >= leaf_cost))
        return false;
      float lo = axis(cmn, best_axis), hi = axis(cmx, best_axis);
      float scale = float(kBins) / (hi - lo);
      Prim* it = std::partition(prims + first, prims + first + count, [&](const Prim& p) {
        return std::min(kBins - 1, int((axis(p.centroid, best_axis) - lo) * scale)) <= best_split;
      });
      mid = uint32_t(it - prims);
      if ((mid == first) || (mid == first + count))
        mid = first + count / 2;
    } else if (count <= kMaxLeaf) {
      return false;
    }
    return true;
  }

  uint32_t subdivide(uint32_t first, uint32_t count, uint32_t depth) {
    uint32_t index = uint32_t(nodes.size());
    nodes.emplace_back();
    max_depth = std::max(max_depth, depth);
    f3 mn, mx;
    uint32_t mid = 0;
    const bool inner = split(first, count, mn, mx, mid);
    nodes[index].bmin = mn, nodes[index].bmax = mx, nodes[index].first = first, nodes[index].count = count;
    if (inner == false)
      return index;
    uint32_t l = subdivide(first, mid - first, depth + 1);
    uint32_t r = subdivide(mid, first + count - mid, depth + 1);
    nodes[index].left = int32_t(l), nodes[index].right = int32_t(r), nodes[index].count = 0;
    return index;
  }

  // The same tree by tasks: a range above `grain` primitives is split here, its left half built by another thread, and the two
  // subtrees are appended behind their parent. Ranges are disjoint, so the in-place partitions do not meet; every split sees the
  // primitives in the order the sequential build would have left them, so the tree is the sequential one (the node NUMBERING
  // differs, which nothing downstream reads: the BVH4 is numbered breadth first from the structure).
  struct Subtree {
    std::vector<TmpNode> nodes;
    uint32_t max_depth = 0;
  };
  static Subtree build_range(Prim* prims, float traversal_cost, uint32_t first, uint32_t count, uint32_t depth, uint32_t grain) {
    Builder b;
    b.prims = prims;
    b.traversal_cost = traversal_cost;
    if (count <= grain) {
      b.nodes.reserve(size_t(count));
      b.subdivide(first, count, depth);
      return {std::move(b.nodes), b.max_depth};
    }
    TmpNode root;
    uint32_t mid = 0;
    const bool inner = b.split(first, count, root.bmin, root.bmax, mid);
    root.first = first, root.count = count;
    Subtree out;
    out.max_depth = depth;
    if (inner == false) {
      out.nodes.push_back(root);
      return out;
    }
    std::future<Subtree> left_task = std::async(std::launch::async, build_range, prims, traversal_cost, first, mid - first, depth + 1u, grain);
    Subtree right = build_range(prims, traversal_cost, mid, first + count - mid, depth + 1u, grain);
    Subtree left = left_task.get();
    root.count = 0;
    root.left = 1, root.right = int32_t(1u + left.nodes.size());
    out.nodes.reserve(1u + left.nodes.size() + right.nodes.size());
    out.nodes.push_back(root);
    for (const Subtree* part : {&left, &right}) {
      const int32_t offset = int32_t(out.nodes.size());
      for (TmpNode node : part->nodes) {
        if (node.count == 0)
          node.left += offset, node.right += offset;
        out.nodes.push_back(node);
      }
    }
    out.max_depth = std::max(left.max_depth, right.max_depth);
    return out;
  }
};

template <class T>
int upload(DeviceScene& out, const T* src, size_t count, const T*& dst, std::string& error) {
  dst = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    error = "hipMalloc failed (" + std::to_string(bytes) + " bytes)";
    return ETX_HIP_ERROR_HIP;
  }
  ((out.alloc_group == 1) ? out.geometry_allocations : ((out.alloc_group == 2) ? out.image_allocations : out.allocations)).push_back(p);
  if (out.transfer == nullptr) {
    error = "internal: the scene has no host transfer object";
    return ETX_HIP_ERROR_STATE;
  }
  if (count > 0) {
    if (int rc = out.transfer->to_device(p, src, count * sizeof(T), nullptr, error))
      return rc;
  }
  dst = reinterpret_cast<const T*>(p);
  return 0;
}

f3 a3(const etx_abi_float3& v) {
  return {v.x, v.y, v.z};
}

}  // namespace

// Flat-sweep primitives (dev_scene.h FlatPrimInfo): the triangles in traversal order, with every pair that forms a
// parallelogram (two shared corners, fourth corner = a_i + a_j - a_k, same material / flags / winding, no alpha test)
// merged into one primitive based at the corner opposite the shared edge.
void build_flat_prims(const etx_abi_scene* scene, const HostBvh& bvh, std::vector<etxd::BvhTri>& prims, std::vector<etxd::FlatPrimInfo>& infos) {
  using namespace etxd;
  const auto* vertices = reinterpret_cast<const etx_abi_vertex*>(scene->vertices.a);
  const auto* triangles = reinterpret_cast<const etx_abi_triangle*>(scene->triangles.a);
  auto ubits = [](float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
  };
  auto fbits = [](uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  const size_t n = bvh.tris.size();
  std::vector<bool> used(n, false);
  auto corner = [&](uint32_t tri, uint32_t k) { return a3(vertices[triangles[tri].i[k]].pos); };
  auto close = [](const f3& a, const f3& b, float tol) { return (fabsf(a.x - b.x) <= tol) && (fabsf(a.y - b.y) <= tol) && (fabsf(a.z - b.z) <= tol); };
  // weights of the three roles as affine functions (c0, c1 a, c2 b) of the parallelogram coordinates
  const float kFirst[3][3] = {{1.0f, -1.0f, -1.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}};   // a + b <= 1: base corner, corner at a = 1, corner at b = 1
  const float kSecond[3][3] = {{-1.0f, 1.0f, 1.0f}, {1.0f, 0.0f, -1.0f}, {1.0f, -1.0f, 0.0f}}; // a + b > 1: far corner, corner at a = 1, corner at b = 1
  for (size_t ia = 0; ia < n; ++ia) {
    if (used[ia])
      continue;
    used[ia] = true;
    const uint32_t ta = ubits(bvh.tris[ia].v0_index.w);
    const uint32_t flags = ubits(bvh.tris[ia].e1_flags.w);
    const uint32_t material = ubits(bvh.tris[ia].e2_mat.w);
    FlatPrimInfo info = {};
    info.tri_a = ta;
    info.tri_b = kInvalid;
    BvhTri prim = bvh.tris[ia];
    // single triangle: (a, b) are its own barycentrics
    info.ua[1] = 1.0f, info.va[2] = 1.0f;
    if ((flags & kTriAlphaTested) == 0u) {
      const f3 a[3] = {corner(ta, 0), corner(ta, 1), corner(ta, 2)};
      const f3 na = a3(triangles[ta].geo_n);
      float extent = 0.0f;
      for (int k = 0; k < 3; ++k)
        extent = std::max(extent, std::max(fabsf(a[k].x), std::max(fabsf(a[k].y), fabsf(a[k].z))));
      const float tol = 1.0e-6f * std::max(1.0f, extent);
      for (size_t ib = ia + 1; (ib < n) && (info.tri_b == kInvalid); ++ib) {
        if (used[ib] || (ubits(bvh.tris[ib].e1_flags.w) != flags) || (ubits(bvh.tris[ib].e2_mat.w) != material))
          continue;
        const uint32_t tb = ubits(bvh.tris[ib].v0_index.w);
        const f3 nb = a3(triangles[tb].geo_n);
        if (na.x * nb.x + na.y * nb.y + na.z * nb.z < 0.9999f)
          continue;
        const f3 b[3] = {corner(tb, 0), corner(tb, 1), corner(tb, 2)};
        // role of every corner of B: index of the equal corner of A, or -1
        int match[3] = {-1, -1, -1};
        int shared = 0;
        for (int m = 0; m < 3; ++m)
          for (int k = 0; k < 3; ++k)
            if ((match[m] < 0) && close(b[m], a[k], tol)) {
              match[m] = k;
              shared++;
            }
        if (shared != 2)
          continue;
        int far_b = (match[0] < 0) ? 0 : ((match[1] < 0) ? 1 : 2);
        int base_k = 3 - match[(far_b + 1) % 3] - match[(far_b + 2) % 3];  // the corner of A that B does not share
        if ((base_k < 0) || (base_k > 2) || (match[(far_b + 1) % 3] == match[(far_b + 2) % 3]))
          continue;
        const int ki = (base_k + 1) % 3, kj = (base_k + 2) % 3;
        const f3 fourth = {a[ki].x + a[kj].x - a[base_k].x, a[ki].y + a[kj].y - a[base_k].y, a[ki].z + a[kj].z - a[base_k].z};
        if (close(b[far_b], fourth, 4.0f * tol) == false)
          continue;
        // parallelogram: base corner a[base_k], a-axis towards a[ki], b-axis towards a[kj]
        used[ib] = true;
        info.tri_b = tb;
        const f3 e1 = {a[ki].x - a[base_k].x, a[ki].y - a[base_k].y, a[ki].z - a[base_k].z};
        const f3 e2 = {a[kj].x - a[base_k].x, a[kj].y - a[base_k].y, a[kj].z - a[base_k].z};
        prim.v0_index = make_float4(a[base_k].x, a[base_k].y, a[base_k].z, 0.0f);
        prim.e1_flags = make_float4(e1.x, e1.y, e1.z, fbits(flags | kTriQuad));
        prim.e2_mat = make_float4(e2.x, e2.y, e2.z, fbits(material));
        // u = weight of corner 1, v = weight of corner 2 (Embree convention, rt.cxx:352-353)
        auto role_a = [&](int corner_index) { return (corner_index == base_k) ? 0 : ((corner_index == ki) ? 1 : 2); };
        for (int c = 0; c < 3; ++c) {
          info.ua[c] = kFirst[role_a(1)][c];
          info.va[c] = kFirst[role_a(2)][c];
        }
        auto role_b = [&](int corner_index) { return (corner_index == far_b) ? 0 : ((match[corner_index] == ki) ? 1 : 2); };
        for (int c = 0; c < 3; ++c) {
          info.ub[c] = kSecond[role_b(1)][c];
          info.vb[c] = kSecond[role_b(2)][c];
        }
      }
    }
    prim.v0_index.w = fbits(uint32_t(prims.size()));
    prims.push_back(prim);
    infos.push_back(info);
  }
}

void build_bvh(const etx_abi_scene* scene, HostBvh& out, bool keep_bvh2) {
  const auto* vertices = reinterpret_cast<const etx_abi_vertex*>(scene->vertices.a);
  const auto* triangles = reinterpret_cast<const etx_abi_triangle*>(scene->triangles.a);
  const auto* materials = reinterpret_cast<const etx_abi_material*>(scene->materials.a);
  const auto* images = reinterpret_cast<const etx_abi_image*>(scene->images.a);
  uint32_t n = uint32_t(scene->triangles.count);
  out = {};
  if (n == 0) {
    out.root = ~int32_t(0);
    out.root4 = ~int32_t(0);
    return;
  }
  const auto phase_begin = std::chrono::steady_clock::now();
  auto phase = [&](const char* name) {
    if (getenv("ETX_HIP_VERBOSE"))
      fprintf(stderr, "[etx_hip] build_bvh %s: %.1f ms since start\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - phase_begin).count());
  };
  Builder b;
  b.traversal_cost = tuning_knob_f("ETX_HIP_BVH_TRAVERSAL_COST", b.traversal_cost);
  std::vector<Builder::Prim> primitives(n);
  b.prims = primitives.data();
  const uint32_t threads = build_threads();
  parallel_chunks(n, threads, [&](uint32_t begin, uint32_t end) {
    for (uint32_t i = begin; i < end; ++i) {
      f3 p0 = a3(vertices[triangles[i].i[0]].pos), p1 = a3(vertices[triangles[i].i[1]].pos), p2 = a3(vertices[triangles[i].i[2]].pos);
      b.prims[i].bmin = fmin3(p0, fmin3(p1, p2));
      b.prims[i].bmax = fmax3(p0, fmax3(p1, p2));
      b.prims[i].centroid = (b.prims[i].bmin + b.prims[i].bmax) * 0.5f;
      b.prims[i].index = i;
    }
  });
  phase("primitive bounds");
  // large scenes are built by tasks (a million triangles: 0.7 s on one core); ETX_HIP_BVH_BUILD_THREADS=1 keeps one thread
  if ((threads > 1u) && (n >= 32768u)) {
    Builder::Subtree tree = Builder::build_range(b.prims, b.traversal_cost, 0u, n, 1u, std::max(4096u, n / (4u * threads)));
    b.nodes = std::move(tree.nodes);
    b.max_depth = tree.max_depth;
  } else {
    b.nodes.reserve(2 * n);
    b.subdivide(0, n, 1);
  }
  out.depth = b.max_depth;
  phase("binned SAH tree");

  out.tris.resize(n);
  parallel_chunks(n, threads, [&](uint32_t begin, uint32_t end) {
  for (uint32_t i = begin; i < end; ++i) {
    uint32_t ti = b.prims[i].index;
    const etx_abi_triangle& t = triangles[ti];
    f3 p0 = a3(vertices[t.i[0]].pos), p1 = a3(vertices[t.i[1]].pos), p2 = a3(vertices[t.i[2]].pos);
    uint32_t flags = 0;
    if (t.material_index < scene->materials.count) {
      const etx_abi_material& m = materials[t.material_index];
      if (m.cls == ETX_MAT_VOID)
        flags |= kTriVoid;
      if (m.cls == ETX_MAT_BOUNDARY)
        flags |= kTriBoundary;
      bool alpha_image = (m.scattering.image_index != ETX_ABI_INVALID) && (m.scattering.image_index < scene->images.count) &&
                         (images[m.scattering.image_index].options & ETX_IMAGE_HAS_ALPHA);
      if ((m.opacity < 1.0f) || alpha_image)
        flags |= kTriAlphaTested;
    }
    f3 e1 = p1 - p0, e2 = p2 - p0;
    auto bits = [](uint32_t u) {
      float f;
      memcpy(&f, &u, 4);
      return f;
    };
    out.tris[i].v0_index = make_float4(p0.x, p0.y, p0.z, bits(ti));
    out.tris[i].e1_flags = make_float4(e1.x, e1.y, e1.z, bits(flags));
    out.tris[i].e2_mat = make_float4(e2.x, e2.y, e2.z, bits(t.material_index));
  }
  });

  phase("traversal triangles");
  // flatten: inner nodes only; a child reference is an inner index (>= 0) or ~((first << 3) | (count - 1))
  std::vector<int32_t> remap(b.nodes.size(), -1);
  uint32_t inner = 0;
  for (size_t i = 0; i < b.nodes.size(); ++i)
    if (b.nodes[i].count == 0)
      remap[i] = int32_t(inner++);
  auto encode = [&](int32_t tmp_index) -> int32_t {
    const auto& tn = b.nodes[tmp_index];
    if (tn.count == 0)
      return remap[tmp_index];
    return ~int32_t((tn.first << 3) | (tn.count - 1u));
  };
  out.nodes.resize(keep_bvh2 ? inner : 0u);  // the two-wide form is the invariants check's input; the device reads the four-wide one
  for (size_t i = 0; keep_bvh2 && (i < b.nodes.size()); ++i) {
    const auto& tn = b.nodes[i];
    if (tn.count != 0)
      continue;
    const auto& c0 = b.nodes[tn.left];
    const auto& c1 = b.nodes[tn.right];
    BvhNode& dn = out.nodes[remap[i]];
    dn.lo0_hi0x = make_float4(c0.bmin.x, c0.bmin.y, c0.bmin.z, c0.bmax.x);
    dn.hi0yz_lo1xy = make_float4(c0.bmax.y, c0.bmax.z, c1.bmin.x, c1.bmin.y);
    dn.lo1z_hi1 = make_float4(c1.bmin.z, c1.bmax.x, c1.bmax.y, c1.bmax.z);
    dn.child0 = encode(tn.left);
    dn.child1 = encode(tn.right);
    dn.pad0 = dn.pad1 = 0;
  }
  out.root = encode(0);

  phase("BVH2 nodes");
  // BVH2 -> BVH4 (dev_scene.h Bvh4Node): a node adopts its grandchildren, the child with the largest surface area first,
  // until it has four children or only leaves. Nodes are numbered breadth first: the first N nodes are the top of the
  // tree, which the traversal kernels keep in LDS.
  out.nodes4.clear();
  out.depth4 = 0;
  if (b.nodes[0].count != 0) {  // the whole scene is one leaf
    out.root4 = encode(0);
    return;
  }
  struct Pending {
    int32_t tmp_index;  // BVH2 inner node that becomes this BVH4 node
    uint32_t level;
  };
  std::vector<Pending> queue;
  queue.push_back({0, 1u});
  out.root4 = 0;
  for (size_t head = 0; head < queue.size(); ++head) {
    const Pending item = queue[head];
    if (item.level > out.depth4)
      out.level_offsets.push_back(uint32_t(head));  // breadth first: the nodes of a level are consecutive
    out.depth4 = std::max(out.depth4, item.level);
    int32_t kids[4] = {b.nodes[item.tmp_index].left, b.nodes[item.tmp_index].right, -1, -1};
    uint32_t kid_count = 2;
    while (kid_count < 4) {
      int best = -1;
      float best_area = -1.0f;
      for (uint32_t k = 0; k < kid_count; ++k) {
        const auto& kn = b.nodes[kids[k]];
        if (kn.count != 0)
          continue;  // leaf
        const float area = Builder::half_area(kn.bmin, kn.bmax);
        if (area > best_area)
          best_area = area, best = int(k);
      }
      if (best < 0)
        break;
      const int32_t expanded = kids[best];
      kids[best] = b.nodes[expanded].left;
      kids[kid_count++] = b.nodes[expanded].right;
    }
    Bvh4Node node = {};
    float lo[3][4], hi[3][4];
    for (uint32_t k = 0; k < 4; ++k) {
      for (int a = 0; a < 3; ++a)
        lo[a][k] = kMaxFloat, hi[a][k] = -kMaxFloat;
      node.child[k] = kBvhEmptyChild;
      if (k >= kid_count)
        continue;
      const auto& kn = b.nodes[kids[k]];
      lo[0][k] = kn.bmin.x, lo[1][k] = kn.bmin.y, lo[2][k] = kn.bmin.z;
      hi[0][k] = kn.bmax.x, hi[1][k] = kn.bmax.y, hi[2][k] = kn.bmax.z;
      if (kn.count != 0) {
        node.child[k] = ~int32_t((kn.first << 3) | (kn.count - 1u));
      } else {
        node.child[k] = int32_t(queue.size());  // breadth-first index of the child node
        queue.push_back({kids[k], item.level + 1u});
      }
    }
    node.lo_x = make_float4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
    node.lo_y = make_float4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]);
    node.lo_z = make_float4(lo[2][0], lo[2][1], lo[2][2], lo[2][3]);
    node.hi_x = make_float4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]);
    node.hi_y = make_float4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]);
    node.hi_z = make_float4(hi[2][0], hi[2][1], hi[2][2], hi[2][3]);
    out.nodes4.push_back(node);
  }
  out.level_offsets.push_back(uint32_t(out.nodes4.size()));
  // Stack entries the near-child-first traversal can need: descending into one child leaves at most the other children of
  // the node on the stack. Children are numbered after their parents, so one reverse pass resolves the recurrence.
  std::vector<uint32_t> need(out.nodes4.size(), 0u);
  for (size_t i = out.nodes4.size(); i-- > 0;) {
    const Bvh4Node& nd = out.nodes4[i];
    uint32_t kids = 0, deepest = 0;
    for (int k = 0; k < 4; ++k) {
      if (nd.child[k] == kBvhEmptyChild)
        continue;
      kids++;
      if (nd.child[k] >= 0)
        deepest = std::max(deepest, need[size_t(nd.child[k])]);
    }
    need[i] = (kids ? kids - 1u : 0u) + deepest;
  }
  out.stack_need = need.empty() ? 0u : need[0];
  phase("BVH4 collapse and stack bound");
}

// The cube the Morton keys of the device build quantize: the scene's bounding sphere (Scene::bounding_sphere_*, computed by the host
// at commit), or the vertices' box when the scene does not carry one.
void lbvh_cube(const etx_abi_scene* scene, f3& cube_min, float& cube_extent) {
  const f3 center = a3(scene->bounding_sphere_center);
  const float radius = scene->bounding_sphere_radius;
  if ((radius > 0.0f) && std::isfinite(radius) && std::isfinite(center.x) && std::isfinite(center.y) && std::isfinite(center.z)) {
    cube_min = center - mk3(radius), cube_extent = 2.0f * radius;
    return;
  }
  const auto* vertices = reinterpret_cast<const etx_abi_vertex*>(scene->vertices.a);
  f3 lo = mk3(kMaxFloat), hi = mk3(-kMaxFloat);
  for (uint64_t i = 0; i < scene->vertices.count; ++i)
    lo = fmin3(lo, a3(vertices[i].pos)), hi = fmax3(hi, a3(vertices[i].pos));
  cube_min = lo;
  cube_extent = std::max(std::max(hi.x - lo.x, hi.y - lo.y), std::max(hi.z - lo.z, 1.0e-20f));
}

void build_lbvh_host(const etx_abi_scene* scene, HostBvh& out) {
  out = {};
  const uint32_t n = uint32_t(scene->triangles.count);
  if (n <= kLbvhLeafMax) {  // the device path is not used for such scenes (flat sweep, host build)
    build_bvh(scene, out);
    return;
  }
  // a DScene over the HOST arrays: the shared per-element functions only follow its pointers
  DScene view = {};
  view.vertices = reinterpret_cast<const etx_abi_vertex*>(scene->vertices.a);
  view.triangles = reinterpret_cast<const etx_abi_triangle*>(scene->triangles.a);
  view.materials = reinterpret_cast<const etx_abi_material*>(scene->materials.a);
  view.material_count = uint32_t(scene->materials.count);
  std::vector<DImage> image_options(scene->images.count);  // the triangle filter reads DImage::options only
  for (uint64_t i = 0; i < scene->images.count; ++i)
    image_options[i].options = reinterpret_cast<const etx_abi_image*>(scene->images.a)[i].options;
  view.images = image_options.data();
  view.image_count = uint32_t(scene->images.count);
  view.triangle_count = n;
  f3 cube_min;
  float cube_extent = 0.0f;
  lbvh_cube(scene, cube_min, cube_extent);
  std::vector<uint64_t> keys(n);
  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; ++i)
    keys[i] = lbvh_morton_key(view, i, cube_min, 1.0f / cube_extent);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });  // a radix sort is stable
  std::vector<uint64_t> sorted_keys(n);
  out.tris.assign(n, BvhTri{});
  for (uint32_t i = 0; i < n; ++i) {
    sorted_keys[i] = keys[order[i]];
    out.tris[i].v0_index.w = lbvh_bits_float(order[i]);
  }
  for (uint32_t i = 0; i < n; ++i)
    bvh_triangle_update(view, out.tris.data(), i);
  std::vector<LbvhNode> radix(n - 1u);
  for (uint32_t i = 0; i + 1u < n; ++i)
    radix[i] = lbvh_node(sorted_keys.data(), int(n), int(i));
  // boxes of the radix nodes: children before parents (the kernel climbs from the leaves; min / max do not care about the order)
  std::vector<f3> box_lo(n - 1u), box_hi(n - 1u);
  {
    std::vector<uint8_t> done(n - 1u, 0);
    std::vector<uint32_t> pending = {0u};
    while (pending.empty() == false) {
      const uint32_t i = pending.back();
      const LbvhNode nd = radix[i];
      const bool left_ready = (nd.first == nd.split) || done[nd.split], right_ready = (nd.split + 1u == nd.last) || done[nd.split + 1u];
      if (left_ready && right_ready) {
        lbvh_join_children(radix.data(), out.tris.data(), box_lo.data(), box_hi.data(), i);
        done[i] = 1;
        pending.pop_back();
      } else {
        if (left_ready == false)
          pending.push_back(nd.split);
        if (right_ready == false)
          pending.push_back(nd.split + 1u);
      }
    }
  }
  const LbvhBoxes boxes = {out.tris.data(), box_lo.data(), box_hi.data()};
  std::vector<uint32_t> queue = {0u}, next;
  uint32_t base = 0;
  while (queue.empty() == false) {
    out.level_offsets.push_back(base);
    next.clear();
    out.nodes4.resize(base + queue.size());
    for (size_t i = 0; i < queue.size(); ++i) {
      int32_t child[4];
      uint32_t inner[4];
      lbvh_collapse(radix.data(), boxes, queue[i], child, inner);
      for (uint32_t k = 0; k < 4u; ++k) {
        if (inner[k] == kInvalid)
          continue;
        child[k] = int32_t(base + queue.size() + next.size());
        next.push_back(inner[k]);
      }
      Bvh4Node& node = out.nodes4[base + i];
      node = {};
      for (uint32_t k = 0; k < 4u; ++k)
        node.child[k] = child[k];
    }
    base += uint32_t(queue.size());
    queue.swap(next);
  }
  out.level_offsets.push_back(base);
  out.depth4 = uint32_t(out.level_offsets.size()) - 1u;
  out.root4 = 0;
  view.bvh_tris = out.tris.data();
  view.bvh_tri_count = n;
  for (size_t level = out.level_offsets.size(); level-- > 1u;)
    for (uint32_t i = out.level_offsets[level - 1u]; i < out.level_offsets[level]; ++i)
      bvh_refit_node(view, out.nodes4.data(), i);
  out.stack_need = out.nodes4[0].pad[0];
}

// The tree built on the device into the geometry group: d.vertices / triangles / materials / images are on the device already.
// `tris`: the traversal triangle buffer to fill (nullptr: allocate one). The node buffer is sized by the triangle count for the build
// and replaced by one of the exact size afterwards.
int build_lbvh_tables(const etx_abi_scene* scene, DeviceScene& out, DScene& d, BvhTri* tris, hipStream_t stream, std::string& error) {
  const uint32_t n = uint32_t(scene->triangles.count);
  d.triangle_count = n;
  d.material_count = uint32_t(scene->materials.count);  // the triangle filter of the build reads the material and image tables
  d.image_count = uint32_t(scene->images.count);
  Bvh4Node *scratch = nullptr, *nodes = nullptr;
  if (hipMalloc(&scratch, size_t(n) * sizeof(Bvh4Node)) != hipSuccess) {
    error = "hipMalloc failed (" + std::to_string(size_t(n) * sizeof(Bvh4Node)) + " bytes for the BVH build)";
    return ETX_HIP_ERROR_HIP;
  }
  if (tris == nullptr) {
    if (hipMalloc(&tris, size_t(n) * sizeof(BvhTri)) != hipSuccess) {
      (void)hipFree(scratch);
      error = "hipMalloc failed (traversal triangles)";
      return ETX_HIP_ERROR_HIP;
    }
    out.geometry_allocations.push_back(tris);
  }
  f3 cube_min;
  float cube_extent = 0.0f;
  lbvh_cube(scene, cube_min, cube_extent);
  LbvhResult result;
  int rc = lbvh_build_device(stream, d, cube_min, cube_extent, scratch, tris, result, error);
  if ((rc == 0) && (result.stack_need > kMaxStackDepth)) {
    error = "the device-built BVH needs " + std::to_string(result.stack_need) + " traversal stack entries (depth " + std::to_string(result.depth) + "), the device stack holds " +
            std::to_string(kMaxStackDepth) + "; build on the host (etx_hip_set_bvh_builder)";
    rc = ETX_HIP_ERROR_UNSUPPORTED;
  }
  if ((rc == 0) && ((hipMalloc(&nodes, size_t(result.node_count) * sizeof(Bvh4Node)) != hipSuccess) ||
                    (hipMemcpyAsync(nodes, scratch, size_t(result.node_count) * sizeof(Bvh4Node), hipMemcpyDeviceToDevice, stream) != hipSuccess) || (hipStreamSynchronize(stream) != hipSuccess))) {
    error = "hipMalloc / copy of the built BVH failed";
    rc = ETX_HIP_ERROR_HIP;
  }
  (void)hipFree(scratch);
  if (rc) {
    if (nodes != nullptr)
      (void)hipFree(nodes);
    return rc;
  }
  out.geometry_allocations.push_back(nodes);
  d.bvh_nodes = nodes, d.bvh_tris = tris;
  d.bvh_node_count = result.node_count, d.bvh_tri_count = n;
  d.flat_prims = nullptr, d.flat_info = nullptr, d.flat_prim_count = 0u;
  d.bvh_root = result.root, d.bvh_depth = result.depth, d.bvh_stack_need = result.stack_need, d.bvh_flat = 0u;
  out.flat_prims = 0u;
  out.bvh_levels = result.level_offsets;
  out.bvh_depth = result.depth;
  out.bvh_bytes = size_t(result.node_count) * sizeof(Bvh4Node) + size_t(n) * sizeof(BvhTri);
  out.bvh_build_ms = result.milliseconds;
  if (getenv("ETX_HIP_VERBOSE"))
    fprintf(stderr, "[etx_hip] device BVH build: %u triangles -> %u nodes, depth %u, stack %u, %.3f ms\n", n, result.node_count, result.depth, result.stack_need, result.milliseconds);
  return 0;
}

// BVH build + upload (geometry group): the BVH4, the traversal triangles, and for a scene of <= kFlatSweepMaxTriangles the flat-sweep primitives
int build_traversal_tables(const etx_abi_scene* scene, DeviceScene& out, DScene& d, std::string& error) {
  int rc = 0;
  if (out.device_bvh_build && (scene->triangles.count > kFlatSweepMaxTriangles))
    return build_lbvh_tables(scene, out, d, nullptr, nullptr, error);
  HostBvh bvh;
  const auto build_begin = std::chrono::steady_clock::now();
  build_bvh(scene, bvh, /* the two-wide intermediate is the invariants check's input only */ false);
  out.bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - build_begin).count();
  // near-child-first traversal of a four-wide tree pushes at most three children per level
  if (bvh.stack_need > kMaxStackDepth) {
    error = "the BVH needs " + std::to_string(bvh.stack_need) + " traversal stack entries (depth " + std::to_string(bvh.depth4) + "), the device stack holds " + std::to_string(kMaxStackDepth);
    return ETX_HIP_ERROR_UNSUPPORTED;
  }
  if ((rc = upload(out, bvh.nodes4.data(), bvh.nodes4.size(), d.bvh_nodes, error)))
    return rc;
  if ((rc = upload(out, bvh.tris.data(), bvh.tris.size(), d.bvh_tris, error)))
    return rc;
  d.bvh_node_count = uint32_t(bvh.nodes4.size());
  d.bvh_tri_count = uint32_t(bvh.tris.size());
  out.bvh_levels = bvh.level_offsets;
  if (bvh.tris.size() <= kFlatSweepMaxTriangles) {
    std::vector<BvhTri> edge_prims;
    std::vector<FlatPrimInfo> infos;
    build_flat_prims(scene, bvh, edge_prims, infos);
    // (v0, e1, e2) -> plane + the two coordinate rows, in double
    std::vector<FlatPrim> prims(edge_prims.size());
    for (size_t i = 0; i < edge_prims.size(); ++i) {
      const BvhTri& t = edge_prims[i];
      const double v0[3] = {t.v0_index.x, t.v0_index.y, t.v0_index.z}, e1[3] = {t.e1_flags.x, t.e1_flags.y, t.e1_flags.z}, e2[3] = {t.e2_mat.x, t.e2_mat.y, t.e2_mat.z};
      auto cross3 = [](const double a[3], const double b[3], double r[3]) {
        r[0] = a[1] * b[2] - a[2] * b[1], r[1] = a[2] * b[0] - a[0] * b[2], r[2] = a[0] * b[1] - a[1] * b[0];
      };
      auto dot3 = [](const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
      double n[3], u[3], v[3];
      cross3(e1, e2, n);
      const double nn = dot3(n, n);
      cross3(e2, n, u);
      cross3(n, e1, v);
      FlatPrim& p = prims[i];
      p = {};
      if (nn > 0.0) {
        p.plane = make_float4(float(n[0]), float(n[1]), float(n[2]), float(-dot3(n, v0)));
        p.row_a = make_float4(float(u[0] / nn), float(u[1] / nn), float(u[2] / nn), float(-dot3(u, v0) / nn));
        p.row_b = make_float4(float(v[0] / nn), float(v[1] / nn), float(v[2] / nn), float(-dot3(v, v0) / nn));
      }  // a degenerate primitive keeps a zero plane: den = 0 -> never hit
      memcpy(&p.flags, &t.e1_flags.w, 4);
      memcpy(&p.material, &t.e2_mat.w, 4);
      // a medium boundary: the medium a ray is in after crossing the primitive against (`medium_against`) or along (`medium_along`) the plane
      // normal N - Medium::Instance of rt.cxx:569-573 (entering = dot(geo_n, direction) < 0 -> int_medium) with the side test moved to N,
      // which the sweep has in scalar registers (the crossing then needs no gather of triangle and material)
      p.medium_against = p.medium_along = ETX_ABI_INVALID;
      if ((p.flags & kTriBoundary) && (p.material < scene->materials.count)) {
        const etx_abi_material& m = reinterpret_cast<const etx_abi_material*>(scene->materials.a)[p.material];
        const etx_abi_float3& g = reinterpret_cast<const etx_abi_triangle*>(scene->triangles.a)[infos[i].tri_a].geo_n;
        const bool same_side = (n[0] * g.x + n[1] * g.y + n[2] * g.z) >= 0.0;
        p.medium_against = same_side ? m.int_medium : m.ext_medium;
        p.medium_along = same_side ? m.ext_medium : m.int_medium;
      }
    }
    if ((rc = upload(out, prims.data(), prims.size(), d.flat_prims, error)) || (rc = upload(out, infos.data(), infos.size(), d.flat_info, error)))
      return rc;
    d.flat_prim_count = uint32_t(prims.size());
    out.flat_prims = uint32_t(prims.size());
    if (getenv("ETX_HIP_VERBOSE"))
      fprintf(stderr, "[etx_hip] flat sweep: %zu triangles -> %zu primitives\n", bvh.tris.size(), prims.size());
  }
  d.bvh_root = bvh.root4;
  d.bvh_depth = bvh.depth4;
  d.bvh_stack_need = bvh.stack_need;
  d.bvh_flat = (bvh.tris.size() <= kFlatSweepMaxTriangles) ? 1u : 0u;
  if (d.bvh_flat != 0u) {  // bit 1: a primitive is alpha tested (per-candidate random draws in sweep order: the matrix-core sweep does not take such scenes)
    for (const BvhTri& t : bvh.tris) {
      uint32_t flags = 0;
      memcpy(&flags, &t.e1_flags.w, 4);
      if (flags & kTriAlphaTested)
        d.bvh_flat |= 2u;
    }
  }
  if (tuning_knob("ETX_HIP_FORCE_BVH", 0u) != 0u)
    d.bvh_flat = 0u;
  out.bvh_depth = bvh.depth4;
  out.bvh_bytes = bvh.nodes4.size() * sizeof(Bvh4Node) + bvh.tris.size() * sizeof(BvhTri);

  return 0;
}

int build_device_scene(const etx_abi_scene* scene, const etx_abi_camera* camera, DeviceScene& out, std::string& error, bool keep) {
  const DScene kept = out.host_copy;  // keep: the geometry / image pointers and the BVH scalars of the scene in place
  const uint32_t kept_flat_prims = out.flat_prims;
  if (keep)
    out.release_tables();
  else
    out.release();
  out.alloc_group = 0;
  if ((scene == nullptr) || (camera == nullptr)) {
    error = "scene / camera is null";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if ((scene->flags & ETX_SCENE_COMMITTED) == 0) {
    error = "scene is not committed (Integrator::can_run, integrator.hxx:85-87)";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  const bool spectral = (scene->flags & ETX_SCENE_SPECTRAL) != 0;
  out.needs_rgb_response = false;
  out.has_subsurface_cb = false;
  out.noise_threshold = scene->noise_threshold;
  {
    // what the scene IS, cheaply (a checkpoint names the scene it belongs to with this): the material and emitter tables, the scalar
    // part of the media, up to 4096 evenly spaced vertices and triangles
    uint32_t h = 2166136261u;  // FNV-1a
    auto mix = [&h](const void* data, size_t size) {
      const uint8_t* b = static_cast<const uint8_t*>(data);
      for (size_t i = 0; i < size; ++i)
        h = (h ^ b[i]) * 16777619u;
    };
    // field by field: the ABI structs carry explicit pad members and alignment padding whose bytes belong to nobody - a host
    // that builds the same scene in another process may leave other values there, and a checkpoint must still find its scene
    auto mix_u32 = [&mix](uint32_t v) { mix(&v, sizeof(v)); };
    auto mix_f32 = [&mix](float v) { mix(&v, sizeof(v)); };
    mix_u32(uint32_t(scene->materials.count)), mix_u32(uint32_t(scene->triangles.count)), mix_u32(uint32_t(scene->vertices.count));
    for (uint64_t i = 0; i < scene->materials.count; ++i) {
      const etx_abi_material& m = static_cast<const etx_abi_material*>(scene->materials.a)[i];
      for (const etx_abi_spectral_image* si : {&m.reflectance, &m.scattering, &m.emission})
        mix_u32(si->spectrum_index), mix_u32(si->image_index);
      for (const etx_abi_sampled_image* si : {&m.roughness, &m.metalness, &m.transmission})
        mix_f32(si->value.x), mix_f32(si->value.y), mix_f32(si->value.z), mix_f32(si->value.w), mix_u32(si->image_index), mix_u32(si->channel);
      mix_u32(m.subsurface.cls);
      mix_u32(m.cls), mix_u32(m.int_medium), mix_u32(m.ext_medium), mix_u32(m.normal_image_index), mix_u32(m.diffuse_variation), mix_u32(m.two_sided);
      mix_f32(m.normal_scale), mix_f32(m.opacity), mix_f32(m.emission_collimation);
    }
    for (uint64_t i = 0; i < scene->emitter_instances.count; ++i) {
      const etx_abi_emitter& e = static_cast<const etx_abi_emitter*>(scene->emitter_instances.a)[i];
      mix_u32(e.cls), mix_u32(e.profile), mix_u32(e.triangle_index), mix_f32(e.spectrum_weight), mix_f32(e.additional_weight), mix_f32(e.triangle_area);
    }
    for (uint64_t i = 0; i < scene->emitter_profiles.count; ++i) {
      const etx_abi_emitter_profile& e = static_cast<const etx_abi_emitter_profile*>(scene->emitter_profiles.a)[i];
      mix_u32(e.emission.spectrum_index), mix_u32(e.emission.image_index), mix_f32(e.direction.x), mix_f32(e.direction.y), mix_f32(e.direction.z), mix_u32(e.cls);
      mix_f32(e.angular_size), mix_f32(e.equivalent_disk_size), mix_f32(e.angular_size_cosine);
    }
    for (uint64_t i = 0; i < scene->mediums.count; ++i) {
      const etx_abi_medium& m = static_cast<const etx_abi_medium*>(scene->mediums.a)[i];
      mix_f32(m.bounds_min.x), mix_f32(m.bounds_min.y), mix_f32(m.bounds_min.z), mix_f32(m.bounds_max.x), mix_f32(m.bounds_max.y), mix_f32(m.bounds_max.z);
      mix_u32(m.cls), mix_u32(m.enable_explicit_connections), mix_u32(m.absorption_index), mix_u32(m.scattering_index), mix_f32(m.phase_function_g), mix_f32(m.max_sigma);
      mix_u32(m.dimensions.x), mix_u32(m.dimensions.y), mix_u32(m.dimensions.z);
    }
    const uint64_t step = std::max<uint64_t>(1u, scene->vertices.count / 4096u);
    for (uint64_t i = 0; i < scene->vertices.count; i += step)
      mix(static_cast<const etx_abi_vertex*>(scene->vertices.a) + i, sizeof(etx_abi_vertex));  // fourteen floats, no padding
    const uint64_t tri_step = std::max<uint64_t>(1u, scene->triangles.count / 4096u);
    for (uint64_t i = 0; i < scene->triangles.count; i += tri_step) {
      const etx_abi_triangle& t = static_cast<const etx_abi_triangle*>(scene->triangles.a)[i];
      mix_u32(t.i[0]), mix_u32(t.i[1]), mix_u32(t.i[2]), mix_u32(t.material_index);
    }
    out.content_hash = h;
  }
  if ((camera->film_size.x == 0) || (camera->film_size.y == 0)) {
    error = "camera film size is zero";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if (scene->emitter_instances.count == 0) {
    error = "scene has no emitters";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }

  const auto* materials = reinterpret_cast<const etx_abi_material*>(scene->materials.a);
  const auto* spectrums = reinterpret_cast<const etx_abi_spectrum*>(scene->spectrums.a);
  const auto* images = reinterpret_cast<const etx_abi_image*>(scene->images.a);
  const auto* mediums = reinterpret_cast<const etx_abi_medium*>(scene->mediums.a);
  const auto* triangles = reinterpret_cast<const etx_abi_triangle*>(scene->triangles.a);

  if (scene->materials.count >= (1ull << 20u)) {  // the walk queue of the bidirectional integrator keeps (material | events << 20) in one word (kernels_bdpt.hip)
    error = "more than 2^20 materials";
    return ETX_HIP_ERROR_UNSUPPORTED;
  }
  out.has_subsurface = false;
  // only materials that geometry references need a device implementation
  std::vector<bool> used(scene->materials.count, false);
  for (uint64_t i = 0; i < scene->triangles.count; ++i)
    if (triangles[i].material_index < scene->materials.count)
      used[triangles[i].material_index] = true;
  for (uint64_t i = 0; i < scene->materials.count; ++i) {
    if (used[i] == false)
      continue;
    const etx_abi_material& m = materials[i];
    if (bsdf_class_supported(m.cls) == false) {
      error = "material class " + std::to_string(m.cls) + " (material " + std::to_string(i) + ") is not implemented by the device path";
      return ETX_HIP_ERROR_UNSUPPORTED;
    }
    if ((m.cls == ETX_MAT_DIFFUSE) && (m.diffuse_variation > 2u)) {
      error = "diffuse_variation " + std::to_string(m.diffuse_variation) + " is unknown (0 Lambert, 1 microfacet, 2 vMF: bsdf_various.hxx:47-69)";
      return ETX_HIP_ERROR_UNSUPPORTED;
    }
    if (spectral && ((m.scattering.image_index != ETX_ABI_INVALID) || (m.reflectance.image_index != ETX_ABI_INVALID) || (m.emission.image_index != ETX_ABI_INVALID)))
      out.needs_rgb_response = true;  // apply_rgb, scene.hxx:249-260 (etx_hip_upload_rgb_response)
    if (m.subsurface.cls > 2u) {
      error = "subsurface class " + std::to_string(m.subsurface.cls) + " is unknown (1 random walk, 2 Christensen-Burley: material.hxx:36-41)";
      return ETX_HIP_ERROR_UNSUPPORTED;
    }
    if (m.subsurface.cls == 2u)
      out.has_subsurface_cb = true;
    if (m.subsurface.cls != 0u)
      out.has_subsurface = true;
  }
  out.generic_materials = false;
  out.simple_materials = true;
  // Shading group per material (dev_scene.h kShadeGroup*): a path is shaded by the kernel of its hit material's group.
  std::vector<uint8_t> groups(scene->materials.count, uint8_t(kShadeGroupSimple));
  // Bidirectional kernels of MIXED scenes (kernels_bdpt.hip kPartSimple / kPartGeneral): which materials' BSDF calls need the out-of-line library. A surface
  // of a subsurface material is also shaded with the scatter material (the entry vertex, bidirectional.cxx:629-633), so it inherits that material's answer.
  std::vector<uint8_t> general_bsdf(scene->materials.count, uint8_t(0));
  auto simple_class = [](const etx_abi_material& m) {
    const bool constant_roughness = m.roughness.image_index == ETX_ABI_INVALID;
    const bool thin_film = m.thinfilm.max_thickness * m.thinfilm.min_thickness > 0.0f;
    const bool lambert = (m.cls == ETX_MAT_DIFFUSE) && (m.diffuse_variation == 0u);
    const bool mirror_conductor = (m.cls == ETX_MAT_CONDUCTOR) && constant_roughness && (m.roughness.value.x == 0.0f) && (m.roughness.value.y == 0.0f) && (thin_film == false);
    return lambert || (m.cls == ETX_MAT_TRANSLUCENT) || (m.cls == ETX_MAT_MIRROR) || (m.cls == ETX_MAT_BOUNDARY) || (m.cls == ETX_MAT_VOID) || mirror_conductor;
  };
  const bool scatter_simple = (scene->subsurface_scatter_material >= scene->materials.count) || simple_class(materials[scene->subsurface_scatter_material]);
  for (uint64_t i = 0; i < scene->materials.count; ++i) {
    const etx_abi_material& m = materials[i];
    const bool constant_roughness = m.roughness.image_index == ETX_ABI_INVALID;
    const float max_roughness = std::max(m.roughness.value.x, m.roughness.value.y);
    const bool thin_film = m.thinfilm.max_thickness * m.thinfilm.min_thickness > 0.0f;
    general_bsdf[i] = uint8_t(((simple_class(m) == false) || ((m.subsurface.cls != 0u) && (scatter_simple == false))) ? 1 : 0);
    // "generic" = a connectible (non delta) surface that is not a plain Lambert diffuse one: its connections and merges
    // go through the general BSDF kernels. Delta-only: Mirror, Thinfilm, Boundary, Void, roughness-0 Conductor / Dielectric.
    const bool lambert = (m.cls == ETX_MAT_DIFFUSE) && (m.diffuse_variation == 0u);
    const bool always_delta = (m.cls == ETX_MAT_MIRROR) || (m.cls == ETX_MAT_THINFILM) || (m.cls == ETX_MAT_BOUNDARY) || (m.cls == ETX_MAT_VOID) ||
                              (((m.cls == ETX_MAT_CONDUCTOR) || (m.cls == ETX_MAT_DIELECTRIC)) && constant_roughness && (max_roughness <= kDeltaAlphaTreshold));
    // "simple" = the shade kernels need neither the Heitz walk nor a sampler-dependent evaluation (dev_bsdf.h)
    const bool mirror_conductor = (m.cls == ETX_MAT_CONDUCTOR) && constant_roughness && (m.roughness.value.x == 0.0f) && (m.roughness.value.y == 0.0f) && (thin_film == false);
    const bool simple = lambert || (m.cls == ETX_MAT_TRANSLUCENT) || (m.cls == ETX_MAT_MIRROR) || (m.cls == ETX_MAT_BOUNDARY) || (m.cls == ETX_MAT_VOID) || mirror_conductor;
    groups[i] = uint8_t((m.subsurface.cls != 0u) ? kShadeGroupSubsurface : (simple ? kShadeGroupSimple : kShadeGroupGeneral));
    if (used[i] == false)
      continue;
    if ((lambert == false) && (always_delta == false))
      out.generic_materials = true;
    if ((simple == false) || (m.subsurface.cls != 0u))
      out.simple_materials = false;
  }
  if (tuning_knob("ETX_HIP_FORCE_GENERIC_MATERIALS", 0u) != 0u) {  // experiments: every surface through the general kernels
    out.simple_materials = false;
    for (auto& g : groups)
      g = (g == kShadeGroupSimple) ? uint8_t(kShadeGroupGeneral) : g;
    for (auto& g : general_bsdf)
      g = 1;
  }
  // a mixed scene: the bidirectional kernels run their inline instantiation over every item and the out-of-line one over the items of general classes only.
  // Not when the scatter material itself is of a general class (every entry and exit vertex of a walk would be: nothing to split, and two listed vertices
  // per path and round would not fit the lists). ETX_HIP_BDPT_BINNING=0: the round-5 behaviour, for A/B runs.
  out.bdpt_binning = (out.simple_materials == false) && scatter_simple && (tuning_knob("ETX_HIP_BDPT_BINNING", 1u) != 0u);
  // PrincipledBSDF (bsdf_principled.hxx:24-114) evaluates Conductor / Dielectric / Plastic on a modified copy of the
  // material; the three copies are static, so they are appended to the table once (dev_bsdf_ool.h resolve_material)
  std::vector<etx_abi_material> material_table(materials, materials + scene->materials.count);
  std::vector<uint32_t> variants(scene->materials.count, kInvalid);
  for (uint64_t i = 0; i < scene->materials.count; ++i) {
    if (materials[i].cls != ETX_MAT_PRINCIPLED)
      continue;
    variants[i] = uint32_t(material_table.size());
    etx_abi_material conductor = materials[i];  // bsdf_principled.hxx:34-40
    conductor.int_ior.cls = kSpectrumClassConductor;
    conductor.int_ior.eta_index = scene->default_conductor_eta;
    conductor.int_ior.k_index = scene->default_conductor_k;
    conductor.scattering.image_index = ETX_ABI_INVALID;
    conductor.cls = ETX_MAT_CONDUCTOR;
    etx_abi_material dielectric = materials[i];  // :42-52
    dielectric.int_ior.cls = kSpectrumClassDielectric;
    dielectric.int_ior.eta_index = scene->default_dielectric_eta;
    dielectric.int_ior.k_index = ETX_ABI_INVALID;
    dielectric.reflectance.image_index = ETX_ABI_INVALID;
    etx_abi_material plastic = dielectric;
    dielectric.cls = ETX_MAT_DIELECTRIC;
    plastic.cls = ETX_MAT_PLASTIC;
    material_table.push_back(conductor);
    material_table.push_back(dielectric);
    material_table.push_back(plastic);
  }
  variants.resize(material_table.size(), kInvalid);
  groups.resize(material_table.size(), uint8_t(kShadeGroupGeneral));
  out.group_general = out.group_subsurface = false;
  for (uint64_t i = 0; i < scene->materials.count; ++i) {
    if (used[i] || (i == scene->subsurface_exit_material)) {
      out.group_general = out.group_general || (groups[i] == kShadeGroupGeneral);
      out.group_subsurface = out.group_subsurface || (groups[i] == kShadeGroupSubsurface);
    }
  }
  DScene d = {};
  int rc = 0;
  if (keep) {
    if ((scene->vertices.count != kept.vertex_count) || (scene->triangles.count != kept.triangle_count) || (scene->images.count != out.image_table.size()) ||
        (scene->mediums.count != out.density_grids.size())) {
      error = "etx_hip_update_scene: vertex, triangle, image and medium counts must be those of the uploaded scene (" + std::to_string(kept.vertex_count) + " / " +
              std::to_string(kept.triangle_count) + " / " + std::to_string(out.image_table.size()) + " / " + std::to_string(out.density_grids.size()) + "); use etx_hip_upload_scene";
      return ETX_HIP_ERROR_INVALID_ARGUMENT;
    }
    d.vertices = kept.vertices, d.triangles = kept.triangles, d.tri_shade = kept.tri_shade;
  } else {
    out.alloc_group = 1;
    if ((rc = upload(out, reinterpret_cast<const etx_abi_vertex*>(scene->vertices.a), scene->vertices.count, d.vertices, error)))
      return rc;
    if ((rc = upload(out, triangles, scene->triangles.count, d.triangles, error)))
      return rc;
    // the shading records of the triangles (dev_scene.h kTriShadeStride): built on the device from the two tables just uploaded
    {
      void* rows = nullptr;
      const size_t bytes = std::max<size_t>(scene->triangles.count, 1u) * kTriShadeStride * sizeof(float4);
      if (hipMalloc(&rows, bytes) != hipSuccess) {
        error = "hipMalloc failed (" + std::to_string(bytes) + " bytes)";
        return ETX_HIP_ERROR_HIP;
      }
      out.geometry_allocations.push_back(rows);
      d.tri_shade = static_cast<const float4*>(rows);
      DScene tables = {};
      tables.vertices = d.vertices, tables.triangles = d.triangles;
      launch_build_tri_shade(nullptr, tables, static_cast<float4*>(rows), uint32_t(scene->triangles.count));
      if ((hipGetLastError() != hipSuccess) || (hipStreamSynchronize(nullptr) != hipSuccess)) {
        error = "building the triangles' shading records failed on the device";
        return ETX_HIP_ERROR_HIP;
      }
    }
    out.alloc_group = 0;
  }
  if ((rc = upload(out, reinterpret_cast<const uint32_t*>(scene->triangle_to_emitter.a), scene->triangle_to_emitter.count, d.triangle_to_emitter, error)))
    return rc;
  if ((rc = upload(out, material_table.data(), material_table.size(), d.materials, error)) || (rc = upload(out, variants.data(), variants.size(), d.material_variants, error)) ||
      (rc = upload(out, groups.data(), groups.size(), d.material_group, error)) || (rc = upload(out, general_bsdf.data(), general_bsdf.size(), d.material_general_bsdf, error)))
    return rc;
  for (uint64_t i = 0; spectral && (i < scene->emitter_profiles.count); ++i)
    if (reinterpret_cast<const etx_abi_emitter_profile*>(scene->emitter_profiles.a)[i].emission.image_index != ETX_ABI_INVALID)
      out.needs_rgb_response = true;  // image environment maps / textured emitters
  if ((rc = upload(out, reinterpret_cast<const etx_abi_emitter_profile*>(scene->emitter_profiles.a), scene->emitter_profiles.count, d.emitter_profiles, error)))
    return rc;
  if ((rc = upload(out, reinterpret_cast<const etx_abi_emitter*>(scene->emitter_instances.a), scene->emitter_instances.count, d.emitters, error)))
    return rc;
  if ((rc = upload(out, reinterpret_cast<const etx_abi_distribution_entry*>(scene->emitters_distribution.values.a), scene->emitters_distribution.values.count, d.emitter_dist, error)))
    return rc;

  std::vector<float4> rgb(scene->spectrums.count);
  for (uint64_t i = 0; i < scene->spectrums.count; ++i)
    rgb[i] = make_float4(spectrums[i].integrated.x, spectrums[i].integrated.y, spectrums[i].integrated.z, 0.0f);
  if ((rc = upload(out, rgb.data(), rgb.size(), d.spectrum_rgb, error)))
    return rc;
  // spectral mode: the (wavelength, power) tables of every spectrum, flattened (SpectralDistribution::spectral_entries)
  d.spectral = spectral ? 1u : 0u;
  if (spectral) {
    std::vector<float2> entries;
    std::vector<uint2> ranges(scene->spectrums.count);
    for (uint64_t i = 0; i < scene->spectrums.count; ++i) {
      const uint32_t count = std::min<uint32_t>(spectrums[i].entry_count, ETX_ABI_SPECTRUM_MAX_ENTRIES);
      ranges[i] = make_uint2(uint32_t(entries.size()), count);
      for (uint32_t k = 0; k < count; ++k)
        entries.push_back(make_float2(spectrums[i].entries[k].wavelength, spectrums[i].entries[k].power));
    }
    if ((rc = upload(out, entries.data(), entries.size(), d.spectrum_entries, error)) || (rc = upload(out, ranges.data(), ranges.size(), d.spectrum_ranges, error)))
      return rc;
  }

  std::vector<DImage> dimages(scene->images.count);
  if (keep)
    dimages = out.image_table;
  out.alloc_group = 2;
  for (uint64_t i = 0; (keep == false) && (i < scene->images.count); ++i) {
    const etx_abi_image& img = images[i];
    DImage& di = dimages[i];
    di = {};
    size_t pixel_count = size_t(img.isize.x) * img.isize.y;
    std::vector<float4> pixels(pixel_count);
    if (img.format == ETX_IMAGE_FORMAT_RGBA8) {
      const uint8_t* src = reinterpret_cast<const uint8_t*>(img.pixels.a);
      for (size_t k = 0; k < pixel_count; ++k)  // math.hxx:689-691 to_float4(ubyte4)
        pixels[k] = make_float4(src[4 * k] / 255.0f, src[4 * k + 1] / 255.0f, src[4 * k + 2] / 255.0f, src[4 * k + 3] / 255.0f);
    } else if ((img.format == ETX_IMAGE_FORMAT_RGBA32F) && (pixel_count > 0)) {
      memcpy(pixels.data(), img.pixels.a, pixel_count * sizeof(float4));
    } else if (pixel_count > 0) {
      error = "image " + std::to_string(i) + " has pixel format " + std::to_string(img.format) + " (the device path reads RGBA8 and RGBA32F, image.hxx:10-14)";
      return ETX_HIP_ERROR_UNSUPPORTED;
    }
    if ((rc = upload(out, pixels.data(), pixels.size(), di.pixels, error)))
      return rc;
    di.fsize = {img.fsize.x, img.fsize.y};
    di.offset = {img.offset.x, img.offset.y};
    di.scale = {img.scale.x, img.scale.y};
    di.isize_x = img.isize.x, di.isize_y = img.isize.y;
    di.normalization = img.normalization;
    di.options = img.options;
    di.y_count = uint32_t(img.y_distribution.values.count);
    if ((rc = upload(out, reinterpret_cast<const etx_abi_distribution_entry*>(img.y_distribution.values.a), di.y_count, di.y_entries, error)))
      return rc;
    const auto* rows = reinterpret_cast<const etx_abi_distribution*>(img.x_distributions.a);
    uint32_t stride = (img.x_distributions.count > 0) ? uint32_t(rows[0].values.count) : 0u;
    std::vector<etx_abi_distribution_entry> flat(size_t(stride) * img.x_distributions.count);
    for (uint64_t r = 0; r < img.x_distributions.count; ++r) {
      if (rows[r].values.count != stride) {
        error = "image sampling table rows differ in length";
        return ETX_HIP_ERROR_UNSUPPORTED;
      }
      memcpy(flat.data() + r * stride, rows[r].values.a, stride * sizeof(etx_abi_distribution_entry));
    }
    di.x_stride = stride;
    if ((rc = upload(out, flat.data(), flat.size(), di.x_entries, error)))
      return rc;
  }
  out.alloc_group = 0;
  out.image_table = dimages;
  if ((rc = upload(out, dimages.data(), dimages.size(), d.images, error)))
    return rc;

  std::vector<DMedium> dmediums(scene->mediums.count);
  for (uint64_t i = 0; i < scene->mediums.count; ++i) {
    const etx_abi_medium& m = mediums[i];
    DMedium& dm = dmediums[i];
    dm = {};
    dm.derived_color = dm.derived_distances = kInvalid;
    dm.bounds_min = a3(m.bounds_min), dm.bounds_max = a3(m.bounds_max);
    auto resolve = [&](uint32_t idx) {  // scene_medium.hxx:146-158: invalid index => zero
      return ((idx == ETX_ABI_INVALID) || (idx >= scene->spectrums.count)) ? mk3(0.0f) : a3(spectrums[idx].integrated);
    };
    dm.absorption = resolve(m.absorption_index);
    dm.scattering = resolve(m.scattering_index);
    dm.absorption_index = (m.absorption_index < scene->spectrums.count) ? m.absorption_index : kInvalid;
    dm.scattering_index = (m.scattering_index < scene->spectrums.count) ? m.scattering_index : kInvalid;
    dm.cls = m.cls;
    dm.explicit_connections = m.enable_explicit_connections;
    dm.g = m.phase_function_g;
    dm.max_sigma = m.max_sigma;
    dm.dim_x = m.dimensions.x, dm.dim_y = m.dimensions.y, dm.dim_z = m.dimensions.z;
    if (m.cls != 0) {  // heterogeneous: the density grid, normalised to [0, 1] by the host's MediumPool
      const uint64_t cells = uint64_t(m.dimensions.x) * m.dimensions.y * m.dimensions.z;
      if ((m.density.a == nullptr) || (m.density.count < cells) || (cells == 0)) {
        error = "heterogeneous medium " + std::to_string(i) + " has no density grid";
        return ETX_HIP_ERROR_INVALID_ARGUMENT;
      }
      if (keep) {
        dm.density = out.density_grids[i];
        if (dm.density == nullptr) {
          error = "etx_hip_update_scene: medium " + std::to_string(i) + " became heterogeneous; use etx_hip_upload_scene";
          return ETX_HIP_ERROR_INVALID_ARGUMENT;
        }
      } else {
        out.alloc_group = 2;
        rc = upload(out, reinterpret_cast<const float*>(m.density.a), cells, dm.density, error);
        out.alloc_group = 0;
        if (rc)
          return rc;
      }
    }
  }
  if (keep == false) {
    out.density_grids.assign(scene->mediums.count, nullptr);
    for (uint64_t i = 0; i < scene->mediums.count; ++i)
      out.density_grids[i] = dmediums[i].density;
  }
  // Subsurface materials under the bidirectional integrator (bidirectional.cxx:729-790): the walk runs through the material's
  // interior medium, or - without one - through a medium derived from its colour and scattering distances
  // (subsurface::remap_channel, scene_bssrdf_subsurface.hxx:17-44). The derived medium becomes a table entry of its own here, so that
  // vertices inside the object name their medium by index like every other vertex. A TEXTURED colour or distance map makes the coefficients a
  // property of the entry POINT, not of the material: such a material gets kSssMediumDynamic and every walk through it appends a row of its own
  // to (a per-lane copy of) this table at run time (round 6; until then etx_hip_begin refused such scenes for this integrator).
  // Spectral scenes: the entry names the two spectra and the device remaps at the path's wavelength.
  std::vector<uint32_t> sss_medium(material_table.size(), kInvalid);
  out.sss_media_complete = true;
  bool dynamic_media = false;
  auto derive_medium = [&](const etx_abi_material& m, bool allow_dynamic) -> uint32_t {
    const bool textured = (m.scattering.image_index != ETX_ABI_INVALID) || (m.subsurface.image_index != ETX_ABI_INVALID);
    if ((m.scattering.spectrum_index >= scene->spectrums.count) || (m.subsurface.spectrum_index >= scene->spectrums.count))
      return kInvalid;
    if (textured) {  // the coefficients are a property of the entry point: the walk appends its own row at run time (DScene::sss_dynamic_media)
      if (allow_dynamic == false)
        return kInvalid;
      dynamic_media = true;
      return kSssMediumDynamic;
    }
    const f3 color = a3(spectrums[m.scattering.spectrum_index].integrated), distances = a3(spectrums[m.subsurface.spectrum_index].integrated);
    auto remap = [](float colour, float distance, float& extinction, float& scattering) {
      const float a = 1.826052378200f, b = 4.985111943850f + 0.12735595943800f, cc = 1.096861024240f;
      const float dd = 0.496310210422f, e = 4.231902997010f + 0.00310603949088f, f = 2.406029994080f;
      colour = std::max(0.0f, colour);
      const float blend = powf(colour, 0.25f);
      float albedo = (1.0f - blend) * a * powf(atanf(b * colour), cc) + blend * dd * powf(atanf(e * colour), f);
      albedo = std::min(std::max(albedo, 0.0f), 1.0f - kEpsilon);
      extinction = 1.0f / std::max(distance, 1.0f / 1024.0f);
      scattering = extinction * albedo;
    };
    f3 extinction, scattering;
    remap(color.x, distances.x, extinction.x, scattering.x);
    remap(color.y, distances.y, extinction.y, scattering.y);
    remap(color.z, distances.z, extinction.z, scattering.z);
    DMedium dm = {};
    dm.absorption = extinction - scattering, dm.scattering = scattering;
    dm.absorption_index = dm.scattering_index = kInvalid;
    dm.derived_color = spectral ? m.scattering.spectrum_index : kInvalid;
    dm.derived_distances = spectral ? m.subsurface.spectrum_index : kInvalid;
    dm.cls = 0u, dm.explicit_connections = 0u, dm.g = 0.0f;
    dmediums.push_back(dm);
    return uint32_t(dmediums.size() - 1u);
  };
  bool any_derived = false;
  for (uint64_t i = 0; i < scene->materials.count; ++i) {
    const etx_abi_material& m = material_table[i];
    if ((used[i] == false) || (m.subsurface.cls == 0u))
      continue;
    if (m.int_medium < scene->mediums.count) {
      sss_medium[i] = m.int_medium;
      continue;
    }
    sss_medium[i] = derive_medium(m, true);
    any_derived = true;
    if (sss_medium[i] == kInvalid)
      out.sss_media_complete = false;
  }
  // The ENTRY vertex of such a walk carries a medium of its own: handle_surface (bidirectional.cxx:629-633) swaps material_index for
  // scene.subsurface_scatter_material BEFORE it derives the instance, so the extinction a connection from that vertex is attenuated
  // with comes from the scatter material's parameters (white, and validate_materials' default distances {1, 0.2, 0.04},
  // scene_representation.cxx:270-272), not from the object's. The device keeps that: the scatter material gets an entry too.
  if (any_derived && (scene->subsurface_scatter_material < scene->materials.count)) {
    sss_medium[scene->subsurface_scatter_material] = derive_medium(material_table[scene->subsurface_scatter_material], false);  // the entry vertex' instance stays a table entry
    if (sss_medium[scene->subsurface_scatter_material] == kInvalid)
      out.sss_media_complete = false;
  }
  for (DMedium& dm : dmediums)
    dm.pack_rows();
  d.sss_dynamic_media = dynamic_media ? 1u : 0u;
  out.medium_table_rows = uint32_t(dmediums.size());
  d.dyn_medium_first = d.dyn_medium_capacity = 0u, d.lane_counters = nullptr;  // per lane (host_api.cpp allocate_pools)
  if ((rc = upload(out, dmediums.data(), dmediums.size(), d.mediums, error)) || (rc = upload(out, sss_medium.data(), sss_medium.size(), d.material_sss_medium, error)))
    return rc;

  if (keep && (scene->triangles.count > kFlatSweepMaxTriangles)) {
    // the geometry group stays: traversal tables and the scalars that describe them
    d.bvh_nodes = kept.bvh_nodes, d.bvh_tris = kept.bvh_tris, d.bvh_node_count = kept.bvh_node_count, d.bvh_tri_count = kept.bvh_tri_count;
    d.flat_prims = kept.flat_prims, d.flat_info = kept.flat_info, d.flat_prim_count = kept.flat_prim_count;
    d.bvh_root = kept.bvh_root, d.bvh_depth = kept.bvh_depth, d.bvh_stack_need = kept.bvh_stack_need, d.bvh_flat = kept.bvh_flat;
    out.flat_prims = kept_flat_prims;
  } else {
    // (a scene small enough for the flat sweep is rebuilt in any case: its primitives are pre-transformed on the host)
    if (keep) {
      std::vector<void*> retained;
      for (void* p : out.geometry_allocations) {
        if ((p == kept.vertices) || (p == kept.triangles) || (p == kept.tri_shade))
          retained.push_back(p);
        else
          (void)hipFree(p);
      }
      out.geometry_allocations.swap(retained);
    }
    out.alloc_group = 1;
    rc = build_traversal_tables(scene, out, d, error);
    out.alloc_group = 0;
    if (rc)
      return rc;
  }
  d.boundary_materials = d.normal_mapped_materials = d.textured_materials = 0u;
  for (uint64_t i = 0; i < scene->materials.count; ++i) {
    const etx_abi_material& m = materials[i];
    d.boundary_materials += (m.cls == ETX_MAT_BOUNDARY) ? 1u : 0u;
    const bool normal_map = (m.normal_image_index != ETX_ABI_INVALID) && (m.normal_scale > kEpsilon);
    d.normal_mapped_materials += normal_map ? 1u : 0u;
    const bool textured = normal_map || (m.reflectance.image_index != ETX_ABI_INVALID) || (m.scattering.image_index != ETX_ABI_INVALID) || (m.emission.image_index != ETX_ABI_INVALID) ||
                          (m.roughness.image_index != ETX_ABI_INVALID) || (m.metalness.image_index != ETX_ABI_INVALID) || (m.transmission.image_index != ETX_ABI_INVALID) ||
                          (m.subsurface.image_index != ETX_ABI_INVALID) || (m.thinfilm.thickness_image != ETX_ABI_INVALID);
    d.textured_materials += textured ? 1u : 0u;
  }
  d.vertex_count = uint32_t(scene->vertices.count);
  d.triangle_count = uint32_t(scene->triangles.count);
  d.material_count = uint32_t(scene->materials.count);
  d.emitter_count = uint32_t(scene->emitter_instances.count);
  d.emitter_dist_count = uint32_t(scene->emitters_distribution.values.count);
  d.emitter_dist_total = scene->emitters_distribution.total_weight;
  d.spectrum_count = uint32_t(scene->spectrums.count);
  d.image_count = uint32_t(scene->images.count);
  d.medium_count = uint32_t(scene->mediums.count);
  d.heterogeneous_mediums = 0u;
  for (uint64_t i = 0; i < scene->mediums.count; ++i)
    d.heterogeneous_mediums += (reinterpret_cast<const etx_abi_medium*>(scene->mediums.a)[i].cls != 0) ? 1u : 0u;
  d.env_count = scene->environment_emitters.count;
  memcpy(d.env_emitters, scene->environment_emitters.emitters, sizeof(d.env_emitters));
  d.bounds_center = a3(scene->bounding_sphere_center);
  d.bounds_radius = scene->bounding_sphere_radius;
  d.min_path_length = scene->min_path_length;
  d.max_path_length = scene->max_path_length;
  d.samples = scene->samples;
  d.random_path_termination = scene->random_path_termination;
  d.radiance_clamp = scene->radiance_clamp;
  d.flags = scene->flags;
  d.pixel_sampler_image = scene->pixel_sampler.image_index;
  d.pixel_sampler_radius = scene->pixel_sampler.radius;
  d.subsurface_exit_material = scene->subsurface_exit_material;
  d.subsurface_scatter_material = scene->subsurface_scatter_material;
  d.default_dielectric_eta = scene->default_dielectric_eta;
  d.default_conductor_eta = scene->default_conductor_eta;
  d.default_conductor_k = scene->default_conductor_k;

  DCamera& c = d.camera;
  memcpy(c.view_proj, camera->view_proj, sizeof(c.view_proj));
  c.position = a3(camera->position), c.side = a3(camera->side), c.up = a3(camera->up), c.direction = a3(camera->direction);
  c.tan_half_fov = camera->tan_half_fov, c.aspect = camera->aspect, c.area = camera->area, c.image_plane = camera->image_plane;
  c.film_w = camera->film_size.x, c.film_h = camera->film_size.y, c.cls = camera->cls;
  c.lens_radius = camera->lens_radius, c.focal_distance = camera->focal_distance;
  c.clip_near = camera->clip_near, c.clip_far = camera->clip_far;
  c.lens_image = camera->lens_image, c.medium_index = camera->medium_index;

  const DScene* dev = nullptr;
  if ((rc = upload(out, &d, 1, dev, error)))
    return rc;
  out.device = const_cast<DScene*>(dev);
  d.self = dev;
  out.host_copy = d;
  if ((rc = out.sync_device_copy(error)))
    return rc;
  out.film_w = camera->film_size.x;
  out.film_h = camera->film_size.y;
  return 0;
}

int update_device_geometry(const etx_abi_scene* scene, DeviceScene& out, hipStream_t stream, bool positions_moved, bool rebuild, std::string& error) {
  DScene& d = out.host_copy;
  if ((scene == nullptr) || (scene->vertices.count != d.vertex_count) || (scene->triangles.count != d.triangle_count)) {
    error = "etx_hip_update_scene: the geometry must keep its vertex and triangle counts; use etx_hip_upload_scene";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if (positions_moved) {
    if (out.transfer == nullptr) {
      error = "internal: the scene has no host transfer object";
      return ETX_HIP_ERROR_STATE;
    }
    if (int rc = out.transfer->to_device(const_cast<etx_abi_vertex*>(d.vertices), scene->vertices.a, scene->vertices.count * sizeof(etx_abi_vertex), stream, error))
      return rc;
    if (int rc = out.transfer->to_device(const_cast<etx_abi_triangle*>(d.triangles), scene->triangles.a, scene->triangles.count * sizeof(etx_abi_triangle), stream, error))
      return rc;
  }
  if (positions_moved)
    launch_build_tri_shade(stream, d, const_cast<float4*>(d.tri_shade), d.triangle_count);
  // a scene small enough for the flat sweep had its traversal tables rebuilt from the host scene (build_device_scene)
  if ((d.triangle_count > kFlatSweepMaxTriangles) && rebuild) {
    // a new tree over the moved vertices, built on the device; the traversal triangle buffer is reused, the node buffer replaced
    Bvh4Node* old_nodes = const_cast<Bvh4Node*>(d.bvh_nodes);
    if (int rc = build_lbvh_tables(scene, out, d, const_cast<BvhTri*>(d.bvh_tris), stream, error))
      return rc;
    out.geometry_allocations.erase(std::remove(out.geometry_allocations.begin(), out.geometry_allocations.end(), static_cast<void*>(old_nodes)), out.geometry_allocations.end());
    (void)hipFree(old_nodes);
    return out.sync_device_copy(error);
  }
  if (d.triangle_count > kFlatSweepMaxTriangles) {
    // positions, and the filter flags that follow the material classes (Void, Boundary, alpha test)
    launch_bvh_triangles_update(stream, d, const_cast<BvhTri*>(d.bvh_tris), d.bvh_tri_count);
    for (size_t level = out.bvh_levels.size(); positions_moved && (level-- > 1u);)  // bvh_levels: first node of every level, then the node count
      launch_bvh_refit_level(stream, d, const_cast<Bvh4Node*>(d.bvh_nodes), out.bvh_levels[level - 1u], out.bvh_levels[level] - out.bvh_levels[level - 1u]);
  }
  if ((hipGetLastError() != hipSuccess) || (hipStreamSynchronize(stream) != hipSuccess)) {
    error = "updating the traversal tables failed on the device";
    return ETX_HIP_ERROR_HIP;
  }
  return 0;
}

}  // namespace etxh
