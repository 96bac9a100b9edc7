// dev_bdpt.h - bidirectional path tracing on the wavefront pipeline: CPUBidirectionalImpl
// (sources/etx/rt/integrators/bidirectional.cxx:315-1488) restated per path segment.
//
// The reference keeps explicit vertex arrays per thread (PathData, :273-297). What its MIS weights actually read of a
// vertex is small: position, shading normal, "is a surface", connectible / mis_connectible, pdf.from_prev and
// pdf.history (mis_camera :1059-1068, mis_light :1070-1077, PathVertex::pdf_area :102-128). pdf.history of vertex k is the
// path's running mis_history at the moment k is created (precompute_*_mis :1012-1057 stores the OLD history into the
// previous vertex, then advances), so every vertex record is final when it is written:
//   light pass   every vertex of the emitter path (connectible or not) goes to the light vertex pool, write once; the
//                connection of the vertices of a bounce to the camera (connect_light_to_camera :1380-1428) runs as a kernel
//                over the pool range of that bounce (k_bdpt_connect_camera)
//   camera pass  the path state carries (previous vertex summary, mis_history, pdf_dir); direct emitter hits are resolved
//                in the step (no visibility query involved); every connectible vertex goes to the camera vertex pool with
//                its previous-vertex summary; next event estimation (connect_camera_to_light :1342-1378, k_bdpt_connect_light)
//                and the vertex connections (connect_camera_to_light_path :438-497, k_bdpt_expand_pairs + k_bdpt_connect_pairs)
//                read it there; visibility goes through the shadow queue like every other connection.
// Modes (CPUBidirectionalImpl::Mode :323-330): PathTracing, LightTracing, BDPTFast (the reference's default: no vertex
// connections, product-form weights), BDPTFull. Random-walk subsurface materials: the reference threads the walk's medium vertices through
// the path (:729-818); here such a path leaves the wavefront for the walk queue (kernels_bdpt.hip).
#pragma once

#include "dev_vcm.h"

namespace etxd {

enum : uint32_t { kBdptPathTracing = 0, kBdptLightTracing = 1, kBdptFast = 2, kBdptFull = 3 };  // bidirectional.cxx:323-330

// vertex flags (PathVertex: cls, connectible, mis_connectible, is_surface_interaction)
enum : uint32_t {
  kBvSurface = 1u << 0,         // intersection.triangle_index is valid (surface hit or area-emitter vertex)
  kBvMedium = 1u << 1,          // Class::Medium
  kBvConnectible = 1u << 2,
  kBvMisConnectible = 1u << 3,
  kBvEmitter = 1u << 4,         // Class::Emitter
  kBvScatterMaterial = 1u << 6,     // entry / exit vertex of a subsurface walk: its BSDF is scene.subsurface_scatter_material (handle_surface :612, build_path :859)
  kBvNoCameraConnection = 1u << 5,  // light path medium vertex of a medium without explicit connections: handle_medium skips connect() (:567-569)
  kBvGeneralBsdf = 1u << 7,         // the vertex' material is not of the simple shading group (dev_scene.h): its connections run in the general-BSDF kernels
};

// path flags (meta.w)
enum : uint32_t {
  kBpFirst = 1u << 0,           // no interaction yet (first_interaction of build_path)
  kBpDistantEmitter = 1u << 1,  // light path started on a distant emitter (update_distant_emitter_path_pdfs at the first hit)
  kBpGBuffer = 1u << 2,         // camera path: normal / albedo recorded
  kBpEmitterShift = 8,          // light path: emitter index in the upper bits
};

struct BVtx {  // what MIS needs of a vertex it does not evaluate a BSDF at
  f3 pos, nrm;
  float from_prev, history;
  uint32_t flags, tri;
  ETX_DEV bool surface() const {
    return (flags & kBvSurface) != 0u;
  }
  ETX_DEV bool environment_emitter() const {  // PathVertex::is_environment_emitter :76-78
    return (flags & kBvEmitter) && ((flags & kBvSurface) == 0u);
  }
};

struct BdptState {
  f3 ray_o;
  float ray_tmin;
  f3 ray_d;
  float ray_tmax;
  f3 throughput;
  float eta;
  float pdf_dir, mis_history, aux;  // aux: light path = EmitterSample::pdf_area until the first interaction
  Sampler sampler;
  uint32_t path_size;  // camera_path_size / emitter_path_size
  uint32_t medium, flags, id;
  float wavelength;
  BVtx prev;           // history is filled when the next vertex is created
  uint32_t prev_slot;  // light path: pool slot of the previous vertex
};

ETX_DEV BdptState bdpt_load(const PathSet& set, uint32_t i) {
  BdptState s;
  const float4 a = set.ray_o_tmin[i], b = set.ray_d_tmax[i], c = set.thr_eta[i], d = set.mis[i], e = set.prev_pos[i], f = set.prev_nrm[i];
  const uint4 m = set.meta[i];
  s.ray_o = {a.x, a.y, a.z}, s.ray_tmin = a.w;
  s.ray_d = {b.x, b.y, b.z}, s.ray_tmax = b.w;
  s.throughput = {c.x, c.y, c.z}, s.eta = c.w;
  s.pdf_dir = d.x, s.mis_history = d.y, s.prev.from_prev = d.z, s.aux = d.w;
  s.sampler.seed = m.x, s.sampler.fixed_u = s.sampler.fixed_v = s.sampler.fixed_w = 0.0f;
  s.path_size = m.y, s.medium = m.z, s.flags = m.w;
  s.id = set.path_id[i];
  s.wavelength = set.wavelength[i];
  s.prev.pos = {e.x, e.y, e.z}, s.prev.flags = __float_as_uint(e.w);
  s.prev.nrm = {f.x, f.y, f.z};
  s.prev.tri = kInvalid, s.prev_slot = __float_as_uint(f.w);
  s.prev.history = 0.0f;
  return s;
}

ETX_DEV void bdpt_store(const PathSet& set, uint32_t i, const BdptState& s, uint32_t prev_w) {
  set.ray_o_tmin[i] = mk4(s.ray_o, s.ray_tmin);
  set.ray_d_tmax[i] = mk4(s.ray_d, s.ray_tmax);
  set.thr_eta[i] = mk4(s.throughput, s.eta);
  set.mis[i] = make_float4(s.pdf_dir, s.mis_history, s.prev.from_prev, s.aux);
  set.meta[i] = make_uint4(s.sampler.seed, s.path_size, s.medium, s.flags);
  set.path_id[i] = s.id;
  set.wavelength[i] = s.wavelength;
  set.prev_pos[i] = mk4(s.prev.pos, __uint_as_float(s.prev.flags));
  set.prev_nrm[i] = mk4(s.prev.nrm, __uint_as_float(prev_w));  // light: pool slot of the previous vertex, camera: its triangle
}

ETX_DEV float safe_div(float a, float b) {  // bidirectional.cxx:299-306
  return (b == 0.0f) ? 0.0f : a / b;
}

// PathVertex::convert_solid_angle_pdf_to_area, bidirectional.cxx:224-240
ETX_DEV float bdpt_to_area(float pdf_dir, const f3& from_pos, const BVtx& to) {
  if ((pdf_dir == 0.0f) || to.environment_emitter())
    return pdf_dir;
  f3 w_o = to.pos - from_pos;
  const float d_squared = fmaxf(dot(w_o, w_o), kRayEpsilon * kRayEpsilon);
  const float inv_d_squared = 1.0f / d_squared;
  w_o = w_o * sqrtf(inv_d_squared);
  const float cos_t = to.surface() ? fabsf(dot(w_o, to.nrm)) : 1.0f;
  return cos_t * pdf_dir * inv_d_squared;
}

// A vertex the BSDF / phase function is evaluated at
struct BFull {
  Isect isect;     // surface: the intersection with the vertex' w_i; medium: pos and w_i only
  bool at_medium;
  float g;         // medium anisotropy
  ETX_DEV BVtx summary(float from_prev) const {
    return {isect.pos, isect.nrm, from_prev, 0.0f, at_medium ? uint32_t(kBvMedium) : uint32_t(kBvSurface), at_medium ? kInvalid : isect.tri};
  }
};

// bsdf::reverse_pdf at a vertex of ANY class: the bidirectional MIS weights carry the reverse pdf of every vertex of a path, delta
// ones included (precompute_*_mis, :1012-1057), while the simple-group functions of dev_bsdf.h leave out the classes that are never
// evaluated at a CONNECTIBLE vertex. Here the delta conductor and the mirror answer as scene_bsdf.hxx:94-104 does.
template <bool kSimple>
ETX_DEV float bdpt_reverse_pdf(const DScene& scene, const BsdfData& in_d, const f3& in_w_o, const etx_abi_material& m, Sampler& smp) {
  if (kSimple) {
    BsdfData d = in_d;
    const f3 w_o = -in_d.w_i;
    d.w_i = -in_w_o;
    switch (m.cls) {
      case ETX_MAT_CONDUCTOR:
        return conductor_pdf(scene, d, w_o, m);
      case ETX_MAT_MIRROR: {
        const Frame frame = normal_frame(d);
        return direction_matches(normalize(reflect(d.w_i, frame.nrm)), normalize(w_o)) ? 1.0f : 0.0f;
      }
      default:
        return bsdf_pdf_simple(scene, d, w_o, m);
    }
  }
  return bsdf_reverse_pdf_s<false>(scene, in_d, in_w_o, m, smp);
}

// PathVertex::pdf_area, bidirectional.cxx:102-128: pdf of going prev -> curr -> next, measured as area density at next
template <bool kSimple>
ETX_DEV float bdpt_pdf_area(const DScene& scene, uint32_t source, const f3& prev_pos, const BFull& curr, const BVtx& next, float wavelength, Sampler& smp) {
  f3 w_i = curr.isect.pos - prev_pos;
  float len = dot(w_i, w_i);
  if (len == 0.0f)
    return 0.0f;
  w_i = w_i * (1.0f / sqrtf(len));
  f3 w_o = next.pos - curr.isect.pos;
  len = dot(w_o, w_o);
  if (len == 0.0f)
    return 0.0f;
  w_o = w_o * (1.0f / sqrtf(len));
  float eval_pdf = 0.0f;
  if (curr.at_medium) {
    eval_pdf = phase_function(w_i, w_o, curr.g);
  } else {
    const BsdfData data = make_bsdf_data(curr.isect, w_i, kInvalid, source, wavelength);
    eval_pdf = bsdf_pdf_s<kSimple>(scene, data, w_o, scene.materials[curr.isect.material], smp);
  }
  return bdpt_to_area(eval_pdf, curr.isect.pos, next);
}

// PathVertex::bsdf_in_direction, bidirectional.cxx:242-270
struct BdptBsdf {
  f3 bsdf;
  float pdf;
};
template <bool kSimple>
ETX_DEV BdptBsdf bdpt_bsdf(const DScene& scene, const BFull& v, uint32_t source, const f3& w_o, float wavelength, Sampler& smp) {
  if (v.at_medium) {
    const float p = phase_function(v.isect.w_i, w_o, v.g);
    return {mk3(p), p};
  }
  const BsdfData data = make_bsdf_data(v.isect, v.isect.w_i, kInvalid, source, wavelength);
  BsdfEval e = bsdf_evaluate_s<kSimple>(scene, data, w_o, scene.materials[v.isect.material], smp);
  if (source == kPathLight)
    e.bsdf = e.bsdf * fix_shading_normal(v.isect.geo_n, v.isect.nrm, v.isect.w_i, w_o);
  return {e.bsdf, e.pdf};
}

// PathVertex::emitter_sample_pdf, bidirectional.cxx:178-205
ETX_DEV float bdpt_emitter_sample_pdf(const DScene& s, const etx_abi_emitter& em_inst, const f3& in_direction) {
  const etx_abi_emitter_profile& em = s.emitter_profiles[em_inst.profile];
  const float pdf_discrete = emitter_discrete_pdf(s, em_inst);
  switch (em_inst.cls) {
    case ETX_EMITTER_AREA:
      return pdf_discrete / em_inst.triangle_area;
    case ETX_EMITTER_DIRECTIONAL:
      return direction_matches(in_direction, ld3(em.direction)) ? pdf_discrete : 0.0f;
    default: {  // Environment
      const DImage& img = s.images[em.emission.image_index];
      const f2 uv = direction_to_uv(in_direction, img.offset, img.scale.x);
      const float sin_t = fmaxf(kEpsilon, sinf(uv.y * kPi));
      float image_pdf = 0.0f;
      (void)image_evaluate(img, uv, &image_pdf);
      return pdf_discrete * image_pdf / (2.0f * kPi * kPi * sin_t);
    }
  }
}

// PathVertex::pdf_from_emitter, bidirectional.cxx:154-176. `emitter`: position and normal of the emitter vertex.
ETX_DEV float bdpt_pdf_from_emitter(const DScene& s, uint32_t emitter_index, const f3& emitter_pos, const f3& emitter_nrm, const BVtx& target) {
  const etx_abi_emitter& em_inst = s.emitters[emitter_index];
  if (em_inst.cls == ETX_EMITTER_AREA) {
    const f3 w_o = normalize(target.pos - emitter_pos);
    const float pdf_dir = fmaxf(0.0f, dot(emitter_nrm, w_o)) * kInvPi;  // emitter_evaluate_out_local, scene_emitters.hxx:21-38
    return bdpt_to_area(pdf_dir, emitter_pos, target);
  }
  const f3 w_o = normalize(emitter_pos - target.pos);
  float pdf_area = env_pdf_area(s);  // emitter_evaluate_out_dist, scene_emitters.hxx:107-137
  if (target.surface())
    pdf_area *= fabsf(dot(ld3(s.triangles[target.tri].geo_n), w_o));
  return pdf_area;
}

// mis_camera / mis_light, bidirectional.cxx:1059-1077
ETX_DEV float bdpt_mis_camera(uint32_t camera_path_size, float z_curr_backward, float z_curr_from_prev, float z_prev_backward, const BVtx& z_prev) {
  float acc = 0.0f;
  if (camera_path_size - 1u > 1u) {
    const float r1 = safe_div(z_prev_backward, z_prev.from_prev);
    acc = r1 * (((z_prev.flags & kBvMisConnectible) ? 1.0f : 0.0f) + z_prev.history);
  }
  const float r0 = safe_div(z_curr_backward, z_curr_from_prev);
  return r0 * (((z_prev.flags & kBvConnectible) ? 1.0f : 0.0f) + acc);
}
ETX_DEV float bdpt_mis_light(float y_curr_backward, float y_curr_from_prev, float y_prev_backward, const BVtx& y_prev) {
  const float r1 = safe_div(y_prev_backward, y_prev.from_prev);
  const float acc = r1 * (((y_prev.flags & kBvMisConnectible) ? 1.0f : 0.0f) + y_prev.history);
  const float r0 = safe_div(y_curr_backward, y_curr_from_prev);
  return r0 * (((y_prev.flags & kBvConnectible) ? 1.0f : 0.0f) + acc);
}

ETX_DEV float balance_heuristic(float a, float b, float c) {  // bidirectional.cxx:308-311
  const float denom = a + b + c;
  return (denom == 0.0f) ? 0.0f : a / denom;
}

// precompute_camera_mis / precompute_light_mis, bidirectional.cxx:1012-1057: fixes prev.history, advances the path's
// running history with prev.from_next (just computed by the caller). `curr_connectible`: of the vertex being created.
ETX_DEV void bdpt_advance_history(BdptState& st, float prev_from_next, bool camera, uint32_t mode, bool curr_connectible) {
  st.prev.history = st.mis_history;
  if ((mode == kBdptPathTracing) || (mode == kBdptLightTracing))
    return;
  const float ratio = safe_div(prev_from_next, st.prev.from_prev);
  float accumulated = 0.0f;
  if (mode == kBdptFast) {
    if (camera)  // camera_path_size == 2: "drop backward path, if looking at the scene through the mirror"
      accumulated = (st.path_size == 2u) ? (curr_connectible ? st.mis_history : 0.0f) : st.mis_history * ratio;
    else
      accumulated = st.mis_history * ((st.path_size > 2u) ? ratio : 1.0f);
  } else if ((camera == false) || (st.path_size - 1u > 1u)) {
    accumulated = ratio * (((st.prev.flags & kBvMisConnectible) ? 1.0f : 0.0f) + st.mis_history);
  }
  st.mis_history = accumulated;
}

// ---------------------------------------------------------------------------------------------------------------
// pools

// Light vertex record (LightVertexPool, write once). index_in_path 0 is the emitter vertex itself; its record carries
// pdf.from_next (fixed when vertex 1 is created, read by the BDPTFast weights) in the barycentric slot.
ETX_DEV void bdpt_store_light_vertex(const Pipeline& p, uint32_t idx, uint32_t path, const f3& pos, const f3& nrm, const f3& w_i, const f3& throughput, float from_prev, float history,
  uint32_t flags, uint32_t tri, float bc_u, float bc_v, uint32_t index_in_path, uint32_t path_size, uint32_t medium, uint32_t prev, float wavelength, uint32_t seed) {
  if (idx >= p.lv.capacity) {
    atomicOr(p.counters + kCntOverflow, kOverflowLightVertices);
    return;
  }
  p.lv.pos_dvcm(idx) = mk4(pos, from_prev);
  p.lv.wi_dvc(idx) = mk4(w_i, history);
  p.lv.thr_dvm(idx) = mk4(throughput, __uint_as_float(flags));
  p.lv.nrm_tri(idx) = mk4(nrm, __uint_as_float(tri));
  p.lv.bc_len_med(idx) = make_float4(bc_u, bc_v, __uint_as_float((index_in_path << 16u) | (path_size & 0xffffu)), __uint_as_float(medium));
  p.lv.rec[idx * LightVertexPool::kLvStride + 5] = make_float4(__uint_as_float(prev), wavelength, __uint_as_float(seed), __uint_as_float(path));
  // the path's index list (pipeline.h kBdptRowHeader): one lane owns a path at any time, its vertices arrive in order
  uint32_t* row = reinterpret_cast<uint32_t*>(p.light_path_table) + size_t(path) * p.path_table_entries;
  const bool general = (flags & kBvGeneralBsdf) != 0u;
  const uint32_t entry = idx | (general ? kPathEntryGeneralBit : 0u);
  const uint32_t row_entries = p.path_table_entries - kBdptRowHeader;
  if (row[1] != index_in_path)
    return;  // the list was cut (a chunk could not be had: the iteration is being discarded); it stays as long as what it holds
  if (index_in_path < row_entries) {
    row[kBdptRowHeader + index_in_path] = entry;
  } else {
    const uint32_t position = (index_in_path - row_entries) % kPathChunkEntries;
    uint32_t chunk = row[0];
    if (position == 0u) {  // a new chunk, linked to the one before
      const uint32_t fresh = atomicAdd(p.counters + kCntPathChunks, 1u);
      if (fresh >= p.path_chunk_capacity) {  // cannot happen before the vertex pool itself overflows (host_api.cpp allocate_pools); the list stays as long as its chain
        atomicOr(p.counters + kCntOverflow, kOverflowLightVertices);
        return;
      }
      p.path_chunks[size_t(fresh) * kPathChunkWords] = chunk;
      row[0] = chunk = fresh;
    }
    if (chunk < p.path_chunk_capacity)
      p.path_chunks[size_t(chunk) * kPathChunkWords + 1u + position] = entry;
  }
  row[1] = index_in_path + 1u;
  if (general)
    row[2] += 1u;
}

struct BdptLightVertex {
  BFull full;
  BVtx self;  // from_prev, history, flags of the vertex itself
  f3 throughput;
  uint32_t index_in_path, path_size, medium, prev, seed, path;
  float wavelength;
};

ETX_DEV BVtx bdpt_load_light_summary(const Pipeline& p, uint32_t i) {
  const float4 a = p.lv.pos_dvcm(i), b = p.lv.wi_dvc(i), c = p.lv.thr_dvm(i), d = p.lv.nrm_tri(i);
  return {{a.x, a.y, a.z}, {d.x, d.y, d.z}, a.w, b.w, __float_as_uint(c.w), __float_as_uint(d.w)};
}

ETX_DEV BdptLightVertex bdpt_load_light_vertex(const Pipeline& p, const DScene& scene, uint32_t i) {
  const float4 a = p.lv.pos_dvcm(i), b = p.lv.wi_dvc(i), c = p.lv.thr_dvm(i), d = p.lv.nrm_tri(i), e = p.lv.bc_len_med(i), f = p.lv.rec[i * LightVertexPool::kLvStride + 5];
  BdptLightVertex v;
  v.self = {{a.x, a.y, a.z}, {d.x, d.y, d.z}, a.w, b.w, __float_as_uint(c.w), __float_as_uint(d.w)};
  v.throughput = {c.x, c.y, c.z};
  v.index_in_path = __float_as_uint(e.z) >> 16u, v.path_size = __float_as_uint(e.z) & 0xffffu, v.medium = __float_as_uint(e.w);
  v.prev = __float_as_uint(f.x), v.wavelength = f.y, v.seed = __float_as_uint(f.z), v.path = __float_as_uint(f.w);
  v.full.at_medium = (v.self.flags & kBvMedium) != 0u;
  v.full.g = 0.0f;
  const f3 w_i = {b.x, b.y, b.z};
  if (v.full.at_medium) {
    v.full.isect.pos = v.self.pos, v.full.isect.nrm = mk3(0.0f), v.full.isect.w_i = w_i, v.full.isect.tri = kInvalid;
    v.full.g = (v.medium != kInvalid) ? scene.mediums[v.medium].g : 0.0f;
  } else if (v.self.flags & kBvSurface) {
    v.full.isect = make_intersection(scene, w_i, e.x, e.y, 0.0f, v.self.tri);
    if (v.self.flags & kBvScatterMaterial)
      v.full.isect.material = scene.subsurface_scatter_material;
  } else {
    v.full.isect.pos = v.self.pos, v.full.isect.nrm = v.self.nrm, v.full.isect.w_i = w_i, v.full.isect.tri = kInvalid;
  }
  return v;
}

// Camera vertex record (CameraVertexPool): the connectible vertex z_curr with what its connections read of z_prev.
ETX_DEV void bdpt_store_camera_vertex(const Pipeline& p, uint32_t idx, const BdptState& st, const float4& hit_or_pos, const f3& w_i, uint32_t vertex_medium, const f3& throughput, float from_prev,
  const f3& rnd_fixed, uint32_t seed, bool scatter_material, bool general_bsdf) {
  if (idx >= p.cv_capacity) {
    atomicOr(p.counters + kCntOverflow, kOverflowCameraVertices);
    return;
  }
  p.cv.hit[idx] = hit_or_pos;
  p.cv.wi_medium[idx] = mk4(w_i, __uint_as_float(vertex_medium));
  p.cv.thr_depth[idx] = mk4(throughput, __uint_as_float(st.path_size | (scatter_material ? kCvExitMaterialBit : 0u) | (general_bsdf ? kCvGeneralBsdfBit : 0u)));  // entry / exit vertex of a subsurface walk
  p.cv.mis_pixel[idx] = make_float4(from_prev, st.prev.from_prev, st.prev.history, __uint_as_float(st.id));
  p.cv.seed[idx] = seed;
  p.cv.wavelength[idx] = st.wavelength;
  p.cv.pos_info[idx] = mk4(st.prev.pos, __uint_as_float(st.prev.flags));
  p.cv.nrm_dvm[idx] = mk4(st.prev.nrm, __uint_as_float(st.prev.tri));
  p.cv.fthr_dvcm[idx] = mk4(rnd_fixed, 0.0f);
}

struct BdptCameraVertex {
  BFull full;
  BVtx prev;
  f3 throughput, rnd_fixed;
  float from_prev, wavelength;
  uint32_t path_size, medium, pixel, seed;
};

ETX_DEV BdptCameraVertex bdpt_load_camera_vertex(const Pipeline& p, const DScene& scene, uint32_t i) {
  const float4 h = p.cv.hit[i], w = p.cv.wi_medium[i], t = p.cv.thr_depth[i], m = p.cv.mis_pixel[i], pp = p.cv.pos_info[i], pn = p.cv.nrm_dvm[i], r = p.cv.fthr_dvcm[i];
  BdptCameraVertex v;
  v.throughput = {t.x, t.y, t.z}, v.path_size = __float_as_uint(t.w) & ~(kCvExitMaterialBit | kCvGeneralBsdfBit);
  v.from_prev = m.x;
  v.prev = {{pp.x, pp.y, pp.z}, {pn.x, pn.y, pn.z}, m.y, m.z, __float_as_uint(pp.w), __float_as_uint(pn.w)};
  v.pixel = __float_as_uint(m.w);
  v.medium = __float_as_uint(w.w);
  v.seed = p.cv.seed[i];
  v.wavelength = p.cv.wavelength[i];
  v.rnd_fixed = {r.x, r.y, r.z};
  const f3 w_i = {w.x, w.y, w.z};
  const uint32_t tri = __float_as_uint(h.w);
  v.full.at_medium = tri == kInvalid;
  v.full.g = 0.0f;
  if (v.full.at_medium) {
    v.full.isect.pos = {h.x, h.y, h.z}, v.full.isect.nrm = mk3(0.0f), v.full.isect.w_i = w_i, v.full.isect.tri = kInvalid;
    v.full.g = (v.medium != kInvalid) ? scene.mediums[v.medium].g : 0.0f;
  } else {
    v.full.isect = make_intersection(scene, w_i, h.x, h.y, h.z, tri);
    if (__float_as_uint(t.w) & kCvExitMaterialBit)
      v.full.isect.material = scene.subsurface_scatter_material;
  }
  return v;
}

// local_transmittance, bidirectional.cxx:1430-1438: the segment starts at the vertex' shading position, in the vertex' medium
ETX_DEV f3 bdpt_segment_origin(const DScene& scene, const BFull& v, const f3& towards) {
  if (v.at_medium || (v.isect.tri == kInvalid))
    return v.isect.pos;
  return shading_pos(scene, scene.triangles[v.isect.tri], v.isect.bc, normalize(towards - v.isect.pos));
}

}  // namespace etxd
