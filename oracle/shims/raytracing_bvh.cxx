// ORACLE / TEST INFRASTRUCTURE ONLY - never linked into libetx_hip.so.
//
// etx::Raytracing without Embree.
//
// The reference implements the four ray queries of `struct Raytracing` (sources/etx/rt/rt.hxx:18-43) on
// Embree 4 `rtcIntersect1` with an argument filter callback (sources/etx/rt/rt.cxx:250-278). Embree is an
// external dependency that is neither vendored in /root/reference nor installed in this image (any 4.x,
// unpinned: sources/etx/CMakeLists.txt:73), so this file restates the *filter semantics* of rt.cxx on a plain
// binned-SAH BVH2 with a Moeller-Trumbore triangle test:
//   trace               rt.cxx:428-466   skip Void, stochastic alpha test, keep closest
//   trace_material      rt.cxx:327-371   as trace, restricted to one material id
//   continuous_trace    rt.cxx:373-426   record the first <= N accepted hits of one material
//   trace_transmittance rt.cxx:468-579   collect <= 63 Boundary hits, anything else occludes,
//                                        sort by t, walk media between the hits
// A candidate triangle is handed to the filter when it is hit within [tnear, tfar]; accepting it shrinks tfar
// (rtcIntersect1 semantics). As in the reference, `alpha_test_pass` draws one random number per candidate
// (render/shared/scene_bsdf.hxx:128-144), which makes the per-path random stream depend on traversal order:
// parity with an Embree build is statistical only (SURVEY.md §7 hard part 1) - "parity unpinned" at this boundary.
#include <etx/core/core.hxx>
#include <etx/rt/rt.hxx>
#include <etx/render/host/film.hxx>

#include <algorithm>
#include <cstring>
#include <atomic>
#include <vector>

namespace etx {

namespace {

struct BNode {
  float bmin[3];
  uint32_t left_or_first;  // inner: index of left child (right = left + 1); leaf: first primitive slot
  float bmax[3];
  uint32_t count;  // 0 = inner node
};

struct BTri {
  float3 v0, e1, e2;
  uint32_t index;
};

struct CpuBVH {
  std::vector<BNode> nodes;
  std::vector<BTri> tris;
  // ETX_ORACLE_BVH_ORDER: the order in which the two children of an inner node are visited when the ray enters both.
  // The closest accepted hit does not depend on it; the ORDER in which candidate triangles reach alpha_test_pass (one
  // draw of the path's sampler each, scene_bsdf.hxx:128-144) and their number (a nearer hit found earlier culls more) do.
  // Embree's own order is unknown here; rendering the same scene under several orders measures how far the reference's
  // film moves with it (tests/test_gpu_parity_hi.py).
  enum Order : uint32_t { NearFirst = 0, FarFirst = 1, RandomChild = 2 };
  uint32_t order = NearFirst;

  struct BuildPrim {
    float3 bmin, bmax, centroid;
    uint32_t index;
  };

  static float half_area(const float3& mn, const float3& mx) {
    float3 d = mx - mn;
    return d.x * d.y + d.y * d.z + d.z * d.x;
  }

  void build(const Scene& s) {
    nodes.clear();
    tris.clear();
    uint32_t n = static_cast<uint32_t>(s.triangles.count);
    std::vector<BuildPrim> prims(n);
    for (uint32_t i = 0; i < n; ++i) {
      const auto& t = s.triangles[i];
      const float3& a = s.vertices[t.i[0]].pos;
      const float3& b = s.vertices[t.i[1]].pos;
      const float3& c = s.vertices[t.i[2]].pos;
      prims[i].bmin = min(a, min(b, c));
      prims[i].bmax = max(a, max(b, c));
      prims[i].centroid = (prims[i].bmin + prims[i].bmax) * 0.5f;
      prims[i].index = i;
    }
    nodes.reserve(2u * n + 1u);
    nodes.emplace_back();
    if (n == 0) {
      nodes[0] = {{0, 0, 0}, 0, {0, 0, 0}, 0};
      return;
    }
    subdivide(0, prims, 0, n);
    tris.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
      const auto& t = s.triangles[prims[i].index];
      const float3& a = s.vertices[t.i[0]].pos;
      tris[i] = {a, s.vertices[t.i[1]].pos - a, s.vertices[t.i[2]].pos - a, prims[i].index};
    }
  }

  void subdivide(uint32_t node_index, std::vector<BuildPrim>& prims, uint32_t first, uint32_t count) {
    float3 mn = {kMaxFloat, kMaxFloat, kMaxFloat}, mx = {-kMaxFloat, -kMaxFloat, -kMaxFloat};
    float3 cmn = mn, cmx = mx;
    for (uint32_t i = first; i < first + count; ++i) {
      mn = min(mn, prims[i].bmin);
      mx = max(mx, prims[i].bmax);
      cmn = min(cmn, prims[i].centroid);
      cmx = max(cmx, prims[i].centroid);
    }
    BNode& node = nodes[node_index];
    node.bmin[0] = mn.x, node.bmin[1] = mn.y, node.bmin[2] = mn.z;
    node.bmax[0] = mx.x, node.bmax[1] = mx.y, node.bmax[2] = mx.z;
    node.left_or_first = first;
    node.count = count;
    if (count <= 2)
      return;

    constexpr int kBins = 16;
    float best_cost = kMaxFloat;
    int best_axis = -1;
    int best_split = 0;
    for (int axis = 0; axis < 3; ++axis) {
      float lo = (&cmn.x)[axis], hi = (&cmx.x)[axis];
      if (hi - lo <= 0.0f)
        continue;
      struct Bin {
        float3 mn = {kMaxFloat, kMaxFloat, kMaxFloat}, mx = {-kMaxFloat, -kMaxFloat, -kMaxFloat};
        uint32_t n = 0;
      } bins[kBins];
      float scale = float(kBins) / (hi - lo);
      for (uint32_t i = first; i < first + count; ++i) {
        int b = std::min(kBins - 1, int(((&prims[i].centroid.x)[axis] - lo) * scale));
        bins[b].n++;
        bins[b].mn = min(bins[b].mn, prims[i].bmin);
        bins[b].mx = max(bins[b].mx, prims[i].bmax);
      }
      float left_area[kBins - 1], right_area[kBins - 1];
      uint32_t left_n[kBins - 1], right_n[kBins - 1];
      Bin l, r;
      for (int i = 0; i < kBins - 1; ++i) {
        l.n += bins[i].n, l.mn = min(l.mn, bins[i].mn), l.mx = max(l.mx, bins[i].mx);
        left_n[i] = l.n, left_area[i] = l.n ? half_area(l.mn, l.mx) : 0.0f;
        int j = kBins - 1 - i;
        r.n += bins[j].n, r.mn = min(r.mn, bins[j].mn), r.mx = max(r.mx, bins[j].mx);
        right_n[j - 1] = r.n, right_area[j - 1] = r.n ? half_area(r.mn, r.mx) : 0.0f;
      }
      for (int i = 0; i < kBins - 1; ++i) {
        if ((left_n[i] == 0) || (right_n[i] == 0))
          continue;
        float cost = left_area[i] * float(left_n[i]) + right_area[i] * float(right_n[i]);
        if (cost < best_cost) {
          best_cost = cost, best_axis = axis, best_split = i;
        }
      }
    }

    uint32_t mid = first + count / 2;
    if (best_axis >= 0) {
      float leaf_cost = half_area(mn, mx) * float(count);
      if ((count <= 4) && (best_cost >= leaf_cost))
        return;
      float lo = (&cmn.x)[best_axis], hi = (&cmx.x)[best_axis];
      float scale = float(kBins) / (hi - lo);
      auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const BuildPrim& p) {
        int b = std::min(kBins - 1, int(((&p.centroid.x)[best_axis] - lo) * scale));
        return b <= best_split;
      });
      mid = uint32_t(it - prims.begin());
      if ((mid == first) || (mid == first + count))
        mid = first + count / 2;
    } else if (count <= 4) {
      return;
    }

    uint32_t left = uint32_t(nodes.size());
    nodes.emplace_back();
    nodes.emplace_back();
    nodes[node_index].left_or_first = left;
    nodes[node_index].count = 0;
    subdivide(left, prims, first, mid - first);
    subdivide(left + 1, prims, mid, first + count - mid);
  }

  // filter(triangle_index, u, v, t) -> true to accept the hit (tfar shrinks to t)
  template <class Filter>
  void intersect(const Ray& r, Filter&& filter) const {
    if (tris.empty())
      return;
    const float3 o = r.o, d = r.d;
    const float3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    const float t_near = r.min_t;
    float t_far = r.max_t;

    auto slab = [&](const BNode& n, float& t_enter) {
      float tx0 = (n.bmin[0] - o.x) * inv.x, tx1 = (n.bmax[0] - o.x) * inv.x;
      float ty0 = (n.bmin[1] - o.y) * inv.y, ty1 = (n.bmax[1] - o.y) * inv.y;
      float tz0 = (n.bmin[2] - o.z) * inv.z, tz1 = (n.bmax[2] - o.z) * inv.z;
      float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), t_near));
      float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), t_far));
      t_enter = tmin;
      return tmin <= tmax * 1.0000004f;
    };

    uint32_t stack[64];
    uint32_t sp = 0;
    float t_root = 0.0f;
    if (slab(nodes[0], t_root) == false)
      return;
    stack[sp++] = 0;
    while (sp > 0) {
      const BNode& n = nodes[stack[--sp]];
      if (n.count > 0) {
        for (uint32_t i = n.left_or_first, e = n.left_or_first + n.count; i < e; ++i) {
          const BTri& tr = tris[i];
          float3 p = cross(d, tr.e2);
          float det = dot(tr.e1, p);
          if (det == 0.0f)
            continue;
          float inv_det = 1.0f / det;
          float3 s = o - tr.v0;
          float u = dot(s, p) * inv_det;
          if ((u < 0.0f) || (u > 1.0f))
            continue;
          float3 q = cross(s, tr.e1);
          float v = dot(d, q) * inv_det;
          if ((v < 0.0f) || (u + v > 1.0f))
            continue;
          float t = dot(tr.e2, q) * inv_det;
          if ((t < t_near) || (t > t_far))
            continue;
          if (filter(tr.index, u, v, t)) {
            t_far = t;
          }
        }
        continue;
      }
      float t0 = 0.0f, t1 = 0.0f;
      bool h0 = slab(nodes[n.left_or_first], t0);
      bool h1 = slab(nodes[n.left_or_first + 1u], t1);
      if (h0 && h1) {
        bool left_first = t0 <= t1;
        if (order == FarFirst) {
          left_first = (left_first == false);
        } else if (order == RandomChild) {
          // a fixed pseudo-random choice per (node, ray): no draw from any sampler
          uint32_t h = (n.left_or_first * 0x9E3779B1u) ^ to_uint(d.x) ^ (to_uint(d.y) >> 7u) ^ (to_uint(o.z) << 5u);
          h ^= h >> 15u, h *= 0x2C1B3C6Du, h ^= h >> 12u;
          left_first = (h & 1u) != 0u;
        }
        if (left_first) {
          stack[sp++] = n.left_or_first + 1u;
          stack[sp++] = n.left_or_first;
        } else {
          stack[sp++] = n.left_or_first;
          stack[sp++] = n.left_or_first + 1u;
        }
      } else if (h0) {
        stack[sp++] = n.left_or_first;
      } else if (h1) {
        stack[sp++] = n.left_or_first + 1u;
      }
    }
  }
};

}  // namespace

// ray counters ("counters the oracle must emit")
std::atomic<uint64_t> g_oracle_rays_trace{0};
std::atomic<uint64_t> g_oracle_rays_transmittance{0};
std::atomic<uint64_t> g_oracle_rays_material{0};

// ETX_ORACLE_BVH_DRAWS=opaque_none: alpha_test_pass draws one number of the PATH's sampler per candidate (scene_bsdf.hxx:128-144), also
// for triangles that can never fail the test (opacity 1, no alpha image: `1 <= next()` is false for every draw). In this mode such a
// candidate is tested with a scratch copy of the sampler, so the path's stream no longer depends on how many candidates a query met -
// i.e. on the traversal order of whichever tree is used. The film of the UNMODIFIED integrator (shared light / camera seeds,
// vcm_shared.hxx:312,357) is then unique: identical under every ETX_ORACLE_BVH_ORDER, which is what pins it (tests/test_reference_order_spread.py).
// No reference source is touched; accepted / rejected candidates are the same as without the mode.
static bool g_oracle_opaque_none = false;

static inline bool shim_alpha_test_pass(const Material& mat, const Triangle& tri, const float3& bc, const Scene& scene, Sampler& smp) {
  if (g_oracle_opaque_none && (mat.opacity >= 1.0f)) {
    bool has_alpha = false;
    if (mat.scattering.image_index != kInvalidIndex)
      has_alpha = (scene.images[mat.scattering.image_index].options & Image::HasAlphaChannel) != 0;
    if (has_alpha == false) {
      Sampler scratch = smp;
      return alpha_test_pass(mat, tri, bc, scene, scratch);
    }
  }
  return alpha_test_pass(mat, tri, bc, scene, smp);
}

struct RaytracingImpl {
  TaskScheduler scheduler;
  Film film;
  const Scene* source_scene = nullptr;
  const Camera* source_camera = nullptr;
  CpuBVH* bvh = nullptr;
  bool count_rays = false;
  bool decorrelate = false;  // diagnostic only, see Raytracing::trace
  bool rekey_camera = false; // ETX_ORACLE_DECORRELATE=2, see Raytracing::trace
  bool rekey_second = false; // ETX_ORACLE_DECORRELATE=3, see Raytracing::trace

  RaytracingImpl()
    : film(scheduler) {
    count_rays = getenv("ETX_ORACLE_COUNT_RAYS") != nullptr;
    const char* mode = getenv("ETX_ORACLE_DECORRELATE");
    decorrelate = (mode != nullptr) && (atoi(mode) == 1);
    rekey_camera = (mode != nullptr) && (atoi(mode) == 2);
    rekey_second = (mode != nullptr) && (atoi(mode) == 3);
    const char* draws = getenv("ETX_ORACLE_BVH_DRAWS");
    g_oracle_opaque_none = (draws != nullptr) && (strcmp(draws, "opaque_none") == 0);
  }

  ~RaytracingImpl() {
    delete bvh;
  }
};

ETX_PIMPL_IMPLEMENT(Raytracing, Impl);

Raytracing::Raytracing() {
  ETX_PIMPL_INIT(Raytracing);
}

Raytracing::~Raytracing() {
  ETX_PIMPL_CLEANUP(Raytracing);
}

TaskScheduler& Raytracing::scheduler() {
  return _private->scheduler;
}

void Raytracing::link_scene(const Scene& scene) {
  _private->source_scene = &scene;
}

void Raytracing::link_camera(const Camera& camera) {
  _private->source_camera = &camera;
}

const Camera& Raytracing::camera() const {
  return *_private->source_camera;
}

const Scene& Raytracing::scene() const {
  return *_private->source_scene;
}

const Film& Raytracing::film() const {
  return _private->film;
}

Film& Raytracing::film() {
  return _private->film;
}

void Raytracing::commit_changes() {
  // rt.cxx:58-64: allocate the film for the camera, (re)build the acceleration structure
  _private->film.allocate(_private->source_camera->film_size);
  delete _private->bvh;
  _private->bvh = new CpuBVH();
  if (const char* order = getenv("ETX_ORACLE_BVH_ORDER")) {
    _private->bvh->order = (strcmp(order, "far_first") == 0) ? CpuBVH::FarFirst : ((strcmp(order, "random_child") == 0) ? CpuBVH::RandomChild : CpuBVH::NearFirst);
  }
  _private->bvh->build(*_private->source_scene);
}

bool Raytracing::trace(const Scene& scene, const Ray& r, Intersection& result_intersection, Sampler& smp) const {
  if (_private->count_rays)
    g_oracle_rays_trace.fetch_add(1, std::memory_order_relaxed);
  if (_private->decorrelate) {
    // DIAGNOSTIC (ETX_ORACLE_DECORRELATE=1): the reference seeds light path i and camera path i identically
    // (vcm_shared.hxx:312,357); how far the two streams drift apart depends on how many candidates the traversal
    // hands to alpha_test_pass. Burning a ray-dependent number of extra draws removes any systematic alignment,
    // which tells how much of an oracle-vs-device difference is this correlation and not the estimator.
    uint32_t k = (to_uint(r.o.x) ^ (to_uint(r.d.y) >> 3u) ^ (to_uint(r.o.z) >> 5u)) % 5u;
    for (uint32_t i = 0; i < k; ++i)
      smp.next();
  }
  if (_private->rekey_camera && (_private->source_camera != nullptr) && (_private->source_camera->lens_radius == 0.0f) && (r.o.x == _private->source_camera->position.x) &&
      (r.o.y == _private->source_camera->position.y) && (r.o.z == _private->source_camera->position.z)) {
    // DIAGNOSTIC (ETX_ORACLE_DECORRELATE=2): a primary ray of a pinhole camera = the first segment of a camera sub path.
    // From here on the camera path draws from a stream of its own - exactly what the device does in k_camera_generate
    // (kernels_vcm.hip), only four draws later (pixel jitter and lens sample were taken from the shared stream). Mode 1
    // above merely shifts the shared stream: light and camera path of a pixel still consume the SAME numbers in different
    // roles, which is not independence.
    smp.seed = Sampler::random_seed(smp.seed, 0x43414d45u);
  }
  if (_private->rekey_second && (_private->source_camera != nullptr) && (_private->source_camera->lens_radius == 0.0f)) {
    // DIAGNOSTIC (ETX_ORACLE_DECORRELATE=3): the camera path keeps the shared stream through its FIRST vertex (primary ray, its
    // candidate draws, the six random numbers of that vertex, its connections) and draws from a stream of its own from its second
    // segment on. Tells how much of the correlation of the unmodified reference sits in the first camera vertex - the part a
    // wavefront device could reproduce without serialising its shadow rays (the later alignment of the two streams depends on the
    // candidate draws of every connection ray).
    static thread_local const Sampler* camera_sampler = nullptr;
    const bool primary = (r.o.x == _private->source_camera->position.x) && (r.o.y == _private->source_camera->position.y) && (r.o.z == _private->source_camera->position.z);
    if (primary) {
      camera_sampler = &smp;
    } else if (camera_sampler == &smp) {
      smp.seed = Sampler::random_seed(smp.seed, 0x43414d45u);
      camera_sampler = nullptr;
    }
  }
  IntersectionBase found = {{}, kInvalidIndex, 0.0f};
  _private->bvh->intersect(r, [&](uint32_t triangle_index, float u, float v, float t) {
    const auto& tri = scene.triangles[triangle_index];
    const auto& mat = scene.materials[tri.material_index];
    if (mat.cls == Material::Class::Void)
      return false;
    if (shim_alpha_test_pass(mat, tri, barycentrics({u, v}), scene, smp))
      return false;
    found = {{u, v}, triangle_index, t};
    return true;
  });
  if (found.triangle_index == kInvalidIndex)
    return false;
  result_intersection = make_intersection(scene, r.d, found);
  return true;
}

bool Raytracing::trace_material(const Scene& scene, const Ray& r, const uint32_t material_id, Intersection& result_intersection, Sampler& smp) const {
  if (_private->count_rays)
    g_oracle_rays_material.fetch_add(1, std::memory_order_relaxed);
  IntersectionBase found = {{}, kInvalidIndex, 0.0f};
  _private->bvh->intersect(r, [&](uint32_t triangle_index, float u, float v, float t) {
    const auto& tri = scene.triangles[triangle_index];
    if ((material_id != kInvalidIndex) && (tri.material_index != material_id))
      return false;
    const auto& mat = scene.materials[tri.material_index];
    if (mat.cls == Material::Class::Void)
      return false;
    if (shim_alpha_test_pass(mat, tri, barycentrics({u, v}), scene, smp))
      return false;
    found = {{u, v}, triangle_index, t};
    return true;
  });
  if (found.triangle_index == kInvalidIndex)
    return false;
  result_intersection = make_intersection(scene, r.d, found);
  return true;
}

uint32_t Raytracing::continuous_trace(const Scene& scene, const Ray& r, const ContinousTraceOptions& options, Sampler& smp) const {
  if (_private->count_rays)
    g_oracle_rays_material.fetch_add(1, std::memory_order_relaxed);
  uint32_t count = 0;
  _private->bvh->intersect(r, [&](uint32_t triangle_index, float u, float v, float t) {
    const auto& tri = scene.triangles[triangle_index];
    if ((options.material_id != kInvalidIndex) && (options.material_id != tri.material_index))
      return false;
    const auto& mat = scene.materials[tri.material_index];
    if (mat.cls == Material::Class::Void)
      return false;
    if (shim_alpha_test_pass(mat, tri, barycentrics({u, v}), scene, smp))
      return false;
    if (count < options.max_intersections) {
      options.intersection_buffer[count] = {{u, v}, triangle_index, t};
      count += 1u;
    }
    return count >= options.max_intersections;
  });
  return count;
}

SpectralResponse Raytracing::trace_transmittance(const SpectralQuery spect, const Scene& scene, const float3& p0, const float3& p1, const Medium::Instance& medium,
  Sampler& smp) const {
  if (_private->count_rays)
    g_oracle_rays_transmittance.fetch_add(1, std::memory_order_relaxed);

  constexpr uint32_t kIntersectionBufferSize = 63;
  struct BoundaryHit {
    uint32_t primitive_id;
    float u, v, t;
  } hits[kIntersectionBufferSize + 1u];
  uint32_t hit_count = 0;
  bool occluded = false;

  float3 direction = p1 - p0;
  float t_max = dot(direction, direction);
  if (t_max <= kRayEpsilon)
    return {spect, 1.0f};

  t_max = sqrtf(t_max);
  direction /= t_max;
  t_max -= fmaxf(kRayEpsilon, t_max * kRayEpsilon);

  _private->bvh->intersect(Ray{p0, direction, kRayEpsilon, t_max}, [&](uint32_t triangle_index, float u, float v, float t) {
    const auto& tri = scene.triangles[triangle_index];
    const auto& mat = scene.materials[tri.material_index];
    if (mat.cls == Material::Class::Void)
      return false;
    if (shim_alpha_test_pass(mat, tri, barycentrics({u, v}), scene, smp))
      return false;
    if ((mat.cls != Material::Class::Boundary) || (hit_count + 1u >= kIntersectionBufferSize)) {
      occluded = true;
      return true;
    }
    hits[hit_count++] = {triangle_index, u, v, t};
    return false;
  });

  if (occluded)
    return {spect, 0.0f};

  std::sort(hits, hits + hit_count, [](const BoundaryHit& a, const BoundaryHit& b) {
    return a.t < b.t;
  });
  hits[hit_count++] = {kInvalidIndex, 0.0f, 0.0f, t_max};

  float current_t = 0.0f;
  float3 origin = p0;
  SpectralResponse result = {spect, 1.0f};
  Medium::Instance current_medium = medium;
  for (uint32_t i = 0; i < hit_count; ++i) {
    const auto& h = hits[i];
    if (current_medium.valid()) {
      float dt = fmaxf(0.0f, h.t - current_t);
      if (current_medium.index != kInvalidIndex) {
        result *= medium_transmittance(scene, scene.mediums[current_medium.index], spect, smp, origin, direction, dt);
      } else {
        result *= medium_transmittance(current_medium, dt);
      }
    }
    if (h.primitive_id == kInvalidIndex)
      break;

    const auto& tri = scene.triangles[h.primitive_id];
    const auto& mat = scene.materials[tri.material_index];
    const bool entering_surface = dot(tri.geo_n, direction) < 0.0f;
    current_medium = {.index = entering_surface ? mat.int_medium : mat.ext_medium};
    current_t = h.t;
    origin = lerp_pos(scene.vertices, tri, barycentrics({h.u, h.v}));
  }
  return result;
}

}  // namespace etx
