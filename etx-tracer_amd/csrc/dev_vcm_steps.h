// dev_vcm_steps.h - one segment of a light / camera sub path after the closest-hit query, shared by the wavefront shade
// kernels (kernels_vcm.hip: one launch per bounce) and the tail kernels (kernels_tail.hip: the few long paths that
// survive many bounces loop inside one launch).
#pragma once

#include "dev_vcm.h"

namespace etxd {

ETX_DEV void store_light_vertex(const Pipeline& p, uint32_t idx, const PathState& st, const f3& pos, const f3& nrm, float bc_u, float bc_v, uint32_t tri) {
  if (idx >= p.lv.capacity) {
    atomicOr(p.counters + kCntOverflow, kOverflowLightVertices);
    return;
  }
  p.lv.pos_dvcm(idx) = mk4(pos, st.d_vcm);
  p.lv.wi_dvc(idx) = mk4(st.ray_d, st.d_vc);
  p.lv.thr_dvm(idx) = mk4(st.throughput, st.d_vm);
  p.lv.nrm_tri(idx) = mk4(nrm, __uint_as_float(tri));
  // vcm_connect_to_light_path (vcm_shared.hxx:773-778) derives the connection length from the vertex' INDEX in its
  // light path (delta bounces advance the depth without storing a vertex), the merge uses path_length: keep both.
  uint32_t* row = reinterpret_cast<uint32_t*>(p.light_path_table) + size_t(st.id) * p.path_table_entries;  // head, length, first vertices: one cache line per path
  const uint32_t prev = row[0];
  const uint32_t index_in_path = (prev == kInvalid) ? 0u : ((__float_as_uint(p.lv.bc_len_med(prev).z) >> 16u) + 1u);
  p.lv.bc_len_med(idx) = make_float4(bc_u, bc_v, __uint_as_float((index_in_path << 16u) | (st.depth & 0xffffu)), __uint_as_float(st.medium));
  p.lv.next(idx) = prev;
  p.lv.wavelength(idx) = st.wavelength;
  if (index_in_path + kPathRowHeader < p.path_table_entries)
    row[kPathRowHeader + index_in_path] = idx;
  row[0] = idx;  // the photon bounding box is reduced by k_grid_bbox (kernels_grid.hip)
  row[1] = index_in_path + 1u;
}


// merge_histogram: the iteration merges photons - a vertex k_merge_scatter will sort counts itself into its coarse bucket here
ETX_DEV void store_camera_vertex_m(const Pipeline& p, uint32_t idx, const DScene& scene, const PathState& st, const float4& hit_or_pos, uint32_t seed, const Isect* isect,
  const etx_abi_material& mat, bool exit_material = false, uint32_t use_flags = 0u, bool merge_histogram = false) {  // mat: the vertex' material (unused when isect == nullptr)
  if (idx >= p.cv_capacity) {  // the tail kernel, or more exit points of Christensen-Burley vertices than the pool was sized for
    atomicOr(p.counters + kCntOverflow, kOverflowCameraVertices);
    return;
  }
  p.cv.hit[idx] = hit_or_pos;
  p.cv.wi_medium[idx] = mk4(st.ray_d, __uint_as_float(st.medium));
  p.cv.thr_depth[idx] = mk4(st.throughput, __uint_as_float(st.depth | (exit_material ? kCvExitMaterialBit : 0u)));
  p.cv.mis_pixel[idx] = make_float4(st.d_vcm, st.d_vc, st.d_vm, __uint_as_float(st.id));
  p.cv.seed[idx] = seed;
  p.cv.wavelength[idx] = st.wavelength;
  if (isect == nullptr) {
    p.cv.pos_info[idx] = make_float4(hit_or_pos.x, hit_or_pos.y, hit_or_pos.z, __uint_as_float((st.depth << 8u) | kCvMedium));
    return;
  }
  const bool diffuse = material_is_lambert(mat);
  f3 fthr = st.throughput;
  if (diffuse)
    fthr = fthr * apply_image(scene, mat.scattering, isect->tex, nullptr, st.wavelength) * kInvPi;  // DiffuseBSDF func (bsdf_various.hxx:60-64) x t_camera
  const uint32_t info = (st.depth << 8u) | (diffuse ? kCvDiffuse : 0u) | use_flags;
  p.cv.pos_info[idx] = mk4(isect->pos, __uint_as_float(info));
  p.cv.nrm_dvm[idx] = mk4(isect->nrm, st.d_vm);
  p.cv.fthr_dvcm[idx] = mk4(fthr, st.d_vcm);
  if (merge_histogram) {
    const GridParams& g = *p.grid_params;
    if (merge_candidate(g, scene.max_path_length, info, isect->pos))
      atomicAdd(p.merge_buckets + merge_bucket(g, isect->pos), 1u);
  }
}
ETX_DEV void store_camera_vertex(const Pipeline& p, uint32_t idx, const DScene& scene, const PathState& st, const float4& hit_or_pos, uint32_t seed, const Isect* isect,
  bool exit_material = false, uint32_t use_flags = 0u, bool merge_histogram = false) {
  store_camera_vertex_m(p, idx, scene, st, hit_or_pos, seed, isect, scene.materials[isect ? isect->material : 0u], exit_material, use_flags, merge_histogram);
}


// What happened on a segment (the branches of vcm_light_step / vcm_camera_step)
enum : uint32_t { kEventNone = 0, kEventMedium = 1, kEventSurface = 2, kEventBoundary = 3 };

// vcm_light_step, vcm_shared.hxx:1090-1260: everything after rt.trace for one light sub path segment.
// Returns whether the path continues (state updated in place).
// The reference's medium branch (:1097-1170) and surface branch (:1181-1259) are restated as three phases so that the
// expensive camera connection is a single call site that all lanes of a wave reach together:
//   A  classify the event, draw the randoms, update the MIS quantities at the vertex, sample the BSDF
//   B  store the light vertex, connect it to the camera (same code for medium and surface vertices)
//   C  continue the path (phase function / vcm_next_ray, Russian roulette)
// `slots.get(wanted, counter)` reserves queue / pool slots: the wavefront kernels call this function from
// workgroup-uniform control flow (lanes without a path pass valid = false) and reserve once per workgroup, so the
// reservation points sit outside every data-dependent branch.
// `out` / `out_counter` (wavefront kernels; null in the tail kernels, whose lanes keep their path): the path set a surviving path is appended to.
// The simple shading group takes its three slots - vertex, shadow request, next path set - with ONE reservation at the end of the step
// (Slots::get3) and stores everything after it; the other groups reserve one by one as the step goes (their walks and endpoint requests reserve
// inside data-dependent code anyway) and leave the path to the caller.
template <uint32_t kGroup, class Slots>
ETX_DEV bool light_step(const Pipeline& p, const DScene& scene, const VcmParams& it, PathState& st, const float4& h, bool valid, const Slots& slots, const LaneStack& stack,
  const PathSet* out = nullptr, uint32_t* out_counter = nullptr) {
  constexpr bool kSimple = kGroup == kShadeGroupSimple;      // BSDF classes compiled in (dev_bsdf_ool.h)
  constexpr bool kWalk = kGroup == kShadeGroupSubsurface;   // the subsurface random walk runs inline
  const uint32_t tri = __float_as_uint(h.w);
  const bool found = valid && (tri != kInvalid);
  Isect isect;
  etx_abi_material simple_material;  // kSimple: the hit material's hot fields, read once (load_simple_material); never copied as a whole
  const etx_abi_material& step_material = kSimple ? simple_material : scene.materials[(found && (kSimple == false)) ? scene.triangles[tri].material_index : 0u];
  MediumSample ms;
  ms.sampled_medium_t = 0.0f;
  MediumRows medium_rows = {};  // of st.medium at the start of the segment: free flight, phase function, explicit-connection switch
  uint32_t event = kEventNone;
  if (valid) {
    // the shading point first: its gathers are in flight while the medium is sampled. (Expanding it only for lanes whose free flight reaches the
    // surface - a medium event needs none of it - was measured: fewer gathers, but they start later; 96.9 vs 97.3 Msamples/s on the fog box,
    // 60 vs 62 on one lane, gpurun_out/r4g. make_intersection draws nothing, either order keeps the reference's random numbers.)
    if (found)
      isect = make_intersection(scene, st.ray_d, h.x, h.y, h.z, tri);
    // vcm_try_sampling_medium, vcm_shared.hxx:379-388
    if (st.medium != kInvalid) {
      medium_rows = load_medium_rows(scene.mediums[st.medium]);
      ms = sample_medium_homogeneous(scene, scene.mediums[st.medium], medium_rows, st.wavelength, st.throughput, st.sampler, st.ray_o, st.ray_d, found ? h.z : kMaxFloat);
      st.throughput *= ms.weight;
    }
    // ---- phase A
    if (ms.sampled_medium())
      event = kEventMedium;
    else if (found) {
      if (kSimple)
        load_simple_material(scene, isect.material, simple_material);
      event = vcm_handle_boundary(scene, isect, st, step_material) ? kEventBoundary : kEventSurface;
    }
  }
  const bool scatter_event = (event == kEventMedium) || (event == kEventSurface);  // None: path ends, Boundary: continues as is
  const bool at_medium = event == kEventMedium;
  f2 rnd_bsdf = {0.0f, 0.0f}, rnd_connection = {0.0f, 0.0f}, rnd_support = {0.0f, 0.0f};
  BsdfData bsdf_data;
  BsdfSample bs;
  bool store = false, connect = false;
  bool subsurface_path = false, subsurface_sampled = false;
  Isect ss_isect;
  f3 ss_weight = mk3(0.0f);
  if (scatter_event) {
    // both branches draw the same six numbers (vcm_shared.hxx:1099-1101, 1185-1187)
    rnd_bsdf = st.sampler.next_2d();
    rnd_connection = st.sampler.next_2d();
    rnd_support = st.sampler.next_2d();
    if (at_medium) {
      st.d_vcm *= sqr(st.path_distance + ms.sampled_medium_t);
      st.path_distance = 0.0f;
      store = opt_connect_vertices(it) && (st.depth + 1 <= scene.max_path_length);
      connect = opt_connect_to_camera(it) && medium_rows.explicit_connections && (st.depth + 1 <= scene.max_path_length);
    } else {
      const etx_abi_material& mat = step_material;
      bsdf_data = make_bsdf_data(isect, isect.w_i, st.medium, kPathLight, st.wavelength);
      st.sampler.push_fixed(rnd_bsdf.x, rnd_bsdf.y, rnd_support.x);
      bs = bsdf_sample_s<kSimple>(scene, bsdf_data, mat, st.sampler);
      st.sampler.pop_fixed();
      // vcm_update_light_vcm, vcm_shared.hxx:451-461
      if ((st.depth > 0u) || (st.flags & kPathLocalEmitter))
        st.d_vcm *= sqr(st.path_distance + isect.t);
      float cos_to_prev = fabsf(dot(isect.nrm, -st.ray_d));
      st.d_vcm /= cos_to_prev;
      st.d_vc /= cos_to_prev;
      st.d_vm /= cos_to_prev;
      st.path_distance = 0.0f;
      store = (bs.properties & kSampleDelta) == 0u;  // is_connectible
      connect = store && opt_connect_to_camera(it) && (st.depth + 1 <= scene.max_path_length);
      // vcm_shared.hxx:1198-1200: a diffuse sample on a subsurface material walks through the object
      if (kWalk && (bs.properties & kSampleDiffuse) && (mat.subsurface.cls != 0u)) {
        subsurface_path = true;
        if (mat.subsurface.cls == 2u) {
          // Christensen-Burley: EVERY exit point is connected to the camera with its own weight (vcm_shared.hxx:1210-1224); the
          // exit points are not buffered (dev_sss.h), each request takes its queue slot with a lane-level reservation
          const bool connect_them = connect && (st.depth + 2 <= scene.max_path_length) && (st.depth + 2 >= scene.min_path_length);
          subsurface_sampled = sss_gather_cb(scene, stack, isect, st.sampler, st.wavelength, ss_isect, ss_weight, [&](const Isect& exit_point, const f3& weight) {
            if (connect_them == false)
              return;
            st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
            write_endpoint(p, atomicAdd(p.counters + kCntEndpoints, 1u), make_float4(exit_point.bc.y, exit_point.bc.z, exit_point.t, __uint_as_float(exit_point.tri)), exit_point.w_i,
              st.throughput * weight, true, st.d_vcm, st.d_vc, st.depth, st.medium, st.id, st.wavelength, st.sampler);
            st.sampler.pop_fixed();
          });
          if (subsurface_sampled)
            connect = false;  // done, one request per exit point; a failed gather connects the entry vertex (:1225-1235)
        } else {
          subsurface_sampled = sss_gather_rw(scene, stack, isect, st.sampler, st.wavelength, ss_isect, ss_weight);
        }
        ss_isect.material = scene.subsurface_exit_material;
      }
    }
  }

  // ---- phase C as a function: continue the path (phase function / vcm_next_ray, Russian roulette); true = alive, st advanced
  auto continue_path = [&]() -> bool {
    if (event == kEventBoundary)
      return true;
    if (scatter_event == false)
      return false;
    if (at_medium) {  // vcm_shared.hxx:1143-1169
      f3 w_i = st.ray_d;
      f3 w_o = sample_phase_function(w_i, medium_rows.g, rnd_bsdf);
      float pdf_fwd = phase_function(w_i, w_o, medium_rows.g);
      float pdf_rev = phase_function(w_o, w_i, medium_rows.g);
      st.d_vc = (1.0f / pdf_fwd) * (st.d_vc * pdf_rev + st.d_vcm);
      st.d_vm = (1.0f / pdf_fwd) * (st.d_vm * pdf_rev + 0.0f);
      st.d_vcm = 1.0f / pdf_fwd;
      st.ray_o = ms.pos;
      st.ray_d = w_o;
      st.ray_tmax = kMaxFloat;
      st.ray_tmin = kRayEpsilon;
      st.depth += 1u;
      return (st.depth + 1 <= scene.max_path_length) && random_continue(st.depth, scene.random_path_termination, st.eta, st.sampler, st.throughput);
    }
    if (subsurface_path && (subsurface_sampled == false))
      return false;
    if (kWalk && subsurface_sampled) {  // vcm_shared.hxx:1237-1247: continue from the exit point with a cosine lobe
      st.throughput *= ss_weight;
      bs.w_o = sample_cosine_distribution(rnd_bsdf, ss_isect.nrm, 1.0f);
      bs.pdf = fabsf(dot(bs.w_o, ss_isect.nrm)) / kPi;
      bs.eta = 1.0f;
      bsdf_data = make_bsdf_data(ss_isect, ss_isect.w_i, st.medium, kPathLight, st.wavelength);
      isect = ss_isect;
    }
    if (vcm_next_ray<kSimple>(scene, kPathLight, st, it, isect, bsdf_data, bs, kWalk && subsurface_sampled, (kWalk && subsurface_sampled) ? scene.materials[isect.material] : step_material))
      return st.depth + 1u < scene.max_path_length;
    return false;
  };

  if (kSimple) {  // Lambert / delta surfaces and media: connect inline; ONE reservation for vertex, shadow request and next path set
    ShadowRequest request;
    bool queue = false;
    if (connect) {
      st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
      queue = vcm_connect_to_camera<true>(scene, it, at_medium, &isect, ms.pos, st, request, step_material);
      st.sampler.pop_fixed();
    }
    const PathState at_vertex = st;  // what the vertex record holds; phase C advances st
    const bool alive = continue_path();
    const Slots3 slot = slots.get3(store, p.counters + kCntLightVertices, queue, p.counters + kCntShadow, alive && (out != nullptr), out_counter);
    if (store) {
      if (at_medium)
        store_light_vertex(p, slot.a, at_vertex, ms.pos, mk3(0.0f), 0.0f, 0.0f, kInvalid);
      else
        store_light_vertex(p, slot.a, at_vertex, isect.pos, isect.nrm, isect.bc.y, isect.bc.z, isect.tri);
    }
    if (queue)
      write_shadow(p, slot.b, request);
    if (alive && (out != nullptr))
      store_path(*out, slot.c, st);
    return alive;
  }

  // ---- phase B
  const uint32_t vertex_slot = slots.get(store, p.counters + kCntLightVertices);
  if (store) {
    if (at_medium)
      store_light_vertex(p, vertex_slot, st, ms.pos, mk3(0.0f), 0.0f, 0.0f, kInvalid);
    else
      store_light_vertex(p, vertex_slot, st, isect.pos, isect.nrm, isect.bc.y, isect.bc.z, isect.tri);
  }
  // vcm_shared.hxx:1208-1222: after a walk the connection starts at the exit point, through the exit material, scaled by
  // the walk; when the walk failed the reference connects the entry vertex, then ends the path (:1223-1231, 1249-1251).
  const bool from_exit = kWalk && subsurface_sampled;
  {  // general BSDFs: the evaluation runs in k_connect_endpoints (pipeline.h EndpointQueue)
    connect = connect && (st.depth + 2 <= scene.max_path_length) && (st.depth + 2 >= scene.min_path_length);  // the early-outs of vcm_connect_to_camera
    const uint32_t endpoint_slot = slots.get(connect, p.counters + kCntEndpoints);
    if (connect) {
      st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
      const Isect& from = from_exit ? ss_isect : isect;
      const float4 where = at_medium ? mk4(ms.pos, __uint_as_float(kInvalid)) : make_float4(from.bc.y, from.bc.z, from.t, __uint_as_float(from.tri));
      write_endpoint(p, endpoint_slot, where, at_medium ? st.ray_d : from.w_i, from_exit ? st.throughput * ss_weight : st.throughput, from_exit, st.d_vcm, st.d_vc, st.depth, st.medium, st.id,
        st.wavelength, st.sampler);
      st.sampler.pop_fixed();
    }
  }

  // ---- phase C
  return continue_path();
}

// vcm_camera_step, vcm_shared.hxx:927-1079 after rt.trace, without the vertex connections and the merge: connectible
// vertices go to the camera vertex pool (k_expand_pairs / k_connect_pairs / k_merge consume them), NEE segments go to
// the shadow queue, direct / miss radiance goes straight to the film. Same three-phase shape as light_step.
template <uint32_t kGroup, class Slots>
ETX_DEV bool camera_step(const Pipeline& p, const DScene& scene, const VcmParams& it, PathState& st, const float4& h, bool valid, const Slots& slots, const LaneStack& stack,
  const PathSet* out = nullptr, uint32_t* out_counter = nullptr) {  // out / out_counter: as for light_step
  constexpr bool kSimple = kGroup == kShadeGroupSimple;
  constexpr bool kWalk = kGroup == kShadeGroupSubsurface;
  const uint32_t tri = __float_as_uint(h.w);
  const bool found = valid && (tri != kInvalid);
  Isect isect;
  etx_abi_material simple_material;  // as in light_step
  const etx_abi_material& step_material = kSimple ? simple_material : scene.materials[(found && (kSimple == false)) ? scene.triangles[tri].material_index : 0u];
  MediumSample ms;
  ms.sampled_medium_t = 0.0f;
  MediumRows medium_rows = {};  // of st.medium at the start of the segment: free flight, phase function, explicit-connection switch
  uint32_t event = kEventNone;
  if (valid) {
    if (found)  // before the medium sampling, as in light_step
      isect = make_intersection(scene, st.ray_d, h.x, h.y, h.z, tri);
    if (st.medium != kInvalid) {
      medium_rows = load_medium_rows(scene.mediums[st.medium]);
      ms = sample_medium_homogeneous(scene, scene.mediums[st.medium], medium_rows, st.wavelength, st.throughput, st.sampler, st.ray_o, st.ray_d, found ? h.z : kMaxFloat);
      st.throughput *= ms.weight;
    }
    // ---- phase A
    if (ms.sampled_medium())
      event = kEventMedium;
    else if (found) {
      if (kSimple)
        load_simple_material(scene, isect.material, simple_material);
      event = vcm_handle_boundary(scene, isect, st, step_material) ? kEventBoundary : kEventSurface;
    }
    if (event == kEventNone) {  // vcm_shared.hxx:997-1000
      f3 gathered = vcm_cam_handle_miss(scene, it, st);
      if ((gathered.x != 0.0f) || (gathered.y != 0.0f) || (gathered.z != 0.0f))
        film_add(p, p.camera_sum + film_index(it, st.id), gathered * spectral_film_weight(scene, st.wavelength));
    }
  }
  const bool scatter_event = (event == kEventMedium) || (event == kEventSurface);
  const bool at_medium = event == kEventMedium;

  f2 rnd_bsdf = {0.0f, 0.0f}, rnd_connection = {0.0f, 0.0f}, rnd_support = {0.0f, 0.0f};
  BsdfData bsdf_data;
  BsdfSample bs;
  f3 w_o_medium = mk3(0.0f);
  float pdf_fwd = 0.0f, pdf_rev = 0.0f;
  bool store = false, nee = false;
  bool subsurface_path = false, subsurface_sampled = false, cb_vertex = false;
  Isect ss_isect;
  f3 ss_weight = mk3(0.0f);
  if (scatter_event) {
    // vcm_shared.hxx:936-938, 1013-1015: six numbers from the path's sampler ...
    rnd_bsdf = st.sampler.next_2d();
    rnd_connection = st.sampler.next_2d();
    rnd_support = st.sampler.next_2d();
    // ... replaced by the host's blue-noise samples at the first camera vertex (:941-945, 1018-1022; the sampler has
    // still advanced). pixel_coord = (index % W, index / W) (film.cxx:452-455), sample = iteration, dimensions 0..5
    if ((it.bluenoise != nullptr) && (st.depth == 1u) && (it.iteration < 256u))
      bluenoise_samples(it.bluenoise, st.id % it.film_w, st.id / it.film_w, it.iteration, rnd_bsdf, rnd_connection, rnd_support);
    if (at_medium) {
      st.d_vcm *= sqr(st.path_distance + ms.sampled_medium_t);
      st.path_distance = 0.0f;
      // phase sampling before the explicit connections (vcm_shared.hxx:954-959)
      w_o_medium = sample_phase_function(st.ray_d, medium_rows.g, rnd_bsdf);
      pdf_fwd = phase_function(st.ray_d, w_o_medium, medium_rows.g);
      pdf_rev = phase_function(w_o_medium, st.ray_d, medium_rows.g);
      const bool explicit_connections = medium_rows.explicit_connections && (st.depth + 1 <= scene.max_path_length);
      nee = explicit_connections && opt_connect_to_light(it);
      store = explicit_connections && opt_connect_vertices(it);
    } else {
      const etx_abi_material& mat = step_material;
      bsdf_data = make_bsdf_data(isect, isect.w_i, st.medium, kPathCamera, st.wavelength);
      st.sampler.push_fixed(rnd_bsdf.x, rnd_bsdf.y, rnd_support.x);
      bs = bsdf_sample_s<kSimple>(scene, bsdf_data, mat, st.sampler);
      st.sampler.pop_fixed();
      const bool is_connectible = (bs.properties & kSampleDelta) == 0u;

      // vcm_update_camera_vcm, vcm_shared.hxx:589-595
      float cos_to_prev = fabsf(dot(isect.nrm, -st.ray_d));
      st.d_vcm *= sqr(st.path_distance + isect.t) / cos_to_prev;
      st.d_vc /= cos_to_prev;
      st.d_vm /= cos_to_prev;
      st.path_distance = 0.0f;

      // vcm_handle_direct_hit, vcm_shared.hxx:597-606
      if (opt_direct_hit(it) && (isect.emitter != kInvalid) && (st.depth <= scene.max_path_length) && (st.depth >= scene.min_path_length)) {
        f3 gathered = vcm_get_radiance(scene, scene.emitters[isect.emitter], st, it, isect);
        if ((gathered.x != 0.0f) || (gathered.y != 0.0f) || (gathered.z != 0.0f))
          film_add(p, p.camera_sum + film_index(it, st.id), gathered * spectral_film_weight(scene, st.wavelength));
      }
      nee = is_connectible;
      store = is_connectible && (opt_connect_vertices(it) || (opt_merge_vertices(it) && (st.depth + 1 <= scene.max_path_length)));
      // vcm_shared.hxx:1032-1034
      if (kWalk && (bs.properties & kSampleDiffuse) && (mat.subsurface.cls != 0u)) {
        subsurface_path = true;
        if (mat.subsurface.cls == 2u) {
          // Christensen-Burley (vcm_shared.hxx:1038-1071): EVERY exit point is connected to the light path and to a light with its
          // own weight - a connect-only camera vertex record and an endpoint request per exit point, written as the points are
          // found (dev_sss.h) with lane-level reservations; the photon merge happens once, at the exit point the path continues from
          const bool connect_them = is_connectible;
          const bool light_them = is_connectible && opt_connect_to_light(it) && (st.depth + 1 <= scene.max_path_length) && (st.depth + 1 >= scene.min_path_length);
          subsurface_sampled = sss_gather_cb(scene, stack, isect, st.sampler, st.wavelength, ss_isect, ss_weight, [&](const Isect& exit_point, const f3& weight) {
            Isect e = exit_point;
            e.material = scene.subsurface_exit_material;
            PathState scaled = st;
            scaled.throughput = st.throughput * weight;
            scaled.ray_d = e.w_i;
            const float4 where = make_float4(e.bc.y, e.bc.z, e.t, __uint_as_float(e.tri));
            if (connect_them && opt_connect_vertices(it)) {
              Sampler derived;
              derived.init(st.sampler.seed, 0x51ed270bu ^ e.tri);
              store_camera_vertex(p, atomicAdd(p.counters + kCntCameraVertices, 1u), scene, scaled, where, derived.seed, &e, true, kCvNoMerge);
            }
            if (light_them) {
              st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
              write_endpoint(p, atomicAdd(p.counters + kCntEndpoints, 1u), where, e.w_i, scaled.throughput, true, st.d_vcm, st.d_vc, st.depth, st.medium, st.id, st.wavelength, st.sampler);
              st.sampler.pop_fixed();
            }
          });
          if (subsurface_sampled) {  // a failed gather leaves the entry vertex to be connected and merged as it is (:1048-1054)
            nee = false;
            store = is_connectible && opt_merge_vertices(it) && (st.depth + 1 <= scene.max_path_length);  // the merge-only record
            cb_vertex = true;
          }
        } else {
          subsurface_sampled = sss_gather_rw(scene, stack, isect, st.sampler, st.wavelength, ss_isect, ss_weight);
        }
        ss_isect.material = scene.subsurface_exit_material;
      }
    }
    if (p.debug_flags & 0x100u)  // timing experiments (ETX_HIP_DEBUG_FLAGS)
      nee = false;
    if (p.debug_flags & 0x200u)
      store = false;
  }

#if defined(ETX_HIP_COST_PROBE)
  // cost attribution by duplication (tools/cost_probe.sh): a flag runs one part of the step a second time on copies; the
  // instruction counters of two runs differ by exactly that part. The sink keeps the duplicate alive.
  if (kSimple && (event == kEventSurface)) {
    float sink = 0.0f;
    if (p.debug_flags & 0x400u) {
      const Isect e = make_intersection(scene, st.ray_d, h.y, h.x, h.z, tri);
      sink += e.pos.x + e.nrm.y + e.tan.z + e.btn.x + e.tex.x + __uint_as_float(e.material);
    }
    if ((p.debug_flags & 0x800u) && (st.medium != kInvalid)) {
      Sampler s2 = st.sampler;
      (void)s2.next();
      const MediumSample m2 = sample_medium_homogeneous(scene, scene.mediums[st.medium], st.wavelength, st.throughput, s2, st.ray_o, st.ray_d, h.z);
      sink += m2.pos.x + m2.weight.y + m2.sampled_medium_t;
    }
    if (p.debug_flags & 0x1000u) {
      Sampler s2 = st.sampler;
      s2.push_fixed(rnd_bsdf.y, rnd_bsdf.x, rnd_support.x);
      const BsdfSample b2 = bsdf_sample_s<kSimple>(scene, bsdf_data, scene.materials[isect.material], s2);
      sink += b2.w_o.x + b2.weight.y + b2.pdf;
    }
    if (p.debug_flags & 0x2000u) {
      PathState c = st;
      (void)c.sampler.next();
      const bool a = vcm_next_ray<kSimple>(scene, kPathCamera, c, it, isect, bsdf_data, bs, false);
      sink += c.d_vc + c.d_vm + c.ray_o.x + (a ? 1.0f : 0.0f);
    }
    if (p.debug_flags & 0x4000u) {
      PathState c = st;
      c.sampler.push_fixed(rnd_connection.y, rnd_connection.x, rnd_support.y);
      ShadowRequest r2;
      const bool q2 = vcm_connect_to_light<true>(scene, it, false, &isect, ms.pos, c, film_index(it, st.id), r2, step_material);
      sink += q2 ? r2.value.x + r2.p1.y : 0.0f;
    }
    if (sink == 1.2345e-33f)
      film_add(p, p.camera_sum + film_index(it, st.id), mk3(sink));
  }
#endif
  // ---- phase C as a function: continue the path; true = alive, st advanced
  auto continue_path = [&]() -> bool {
    if (event == kEventBoundary)
      return true;
    if (scatter_event == false)
      return false;
    if (at_medium) {  // vcm_shared.hxx:973-994
      st.d_vc = (1.0f / pdf_fwd) * (st.d_vc * pdf_rev + st.d_vcm);
      st.d_vm = (1.0f / pdf_fwd) * (st.d_vm * pdf_rev + 0.0f);
      st.d_vcm = 1.0f / pdf_fwd;
      st.ray_o = ms.pos;
      st.ray_d = w_o_medium;
      st.ray_tmax = kMaxFloat;
      st.ray_tmin = kRayEpsilon;
      st.depth += 1u;
      return (st.depth + 1 <= scene.max_path_length) && random_continue(st.depth, scene.random_path_termination, st.eta, st.sampler, st.throughput);
    }
    if (subsurface_path && (subsurface_sampled == false))
      return false;
    if (kWalk && subsurface_sampled) {  // vcm_shared.hxx:1049-1061
      st.throughput *= ss_weight;
      bs.w_o = sample_cosine_distribution(rnd_bsdf, ss_isect.nrm, 1.0f);
      bs.pdf = fabsf(dot(bs.w_o, ss_isect.nrm)) / kPi;
      bs.eta = 1.0f;
      bsdf_data = make_bsdf_data(ss_isect, ss_isect.w_i, st.medium, kPathCamera, st.wavelength);
      isect = ss_isect;
    }
    return vcm_next_ray<kSimple>(scene, kPathCamera, st, it, isect, bsdf_data, bs, kWalk && subsurface_sampled, (kWalk && subsurface_sampled) ? scene.materials[isect.material] : step_material);
  };

  if (kSimple) {  // ONE reservation for vertex record, NEE request and next path set (light_step)
    ShadowRequest request;
    bool queue = false;
    if (nee) {
      st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
      queue = vcm_connect_to_light<true>(scene, it, at_medium, &isect, ms.pos, st, film_index(it, st.id), request, step_material);
      st.sampler.pop_fixed();
    }
    const PathState at_vertex = st;
    const bool alive = continue_path();
    const Slots3 slot = slots.get3(store, p.counters + kCntCameraVertices, queue, p.counters + kCntShadow, alive && (out != nullptr), out_counter);
    if (store) {
      Sampler derived;
      derived.init(at_vertex.sampler.seed, 0x51ed270bu);
      if (at_medium)
        store_camera_vertex(p, slot.a, scene, at_vertex, mk4(ms.pos, __uint_as_float(kInvalid)), derived.seed, nullptr);
      else
        store_camera_vertex_m(p, slot.a, scene, at_vertex, h, derived.seed, &isect, step_material, false, 0u, opt_merge_vertices(it));
    }
    if (queue)
      write_shadow(p, slot.b, request);
    if (alive && (out != nullptr))
      store_path(*out, slot.c, st);
    return alive;
  }

  // ---- phase B
  const uint32_t vertex_slot = slots.get(store, p.counters + kCntCameraVertices);
  if (store) {
    Sampler derived;
    derived.init(st.sampler.seed, 0x51ed270bu);
    if (at_medium) {
      store_camera_vertex(p, vertex_slot, scene, st, mk4(ms.pos, __uint_as_float(kInvalid)), derived.seed, nullptr);
    } else if (subsurface_sampled) {
      // vcm_shared.hxx:1036-1046: connections and the merge happen at the exit point, through the exit material, with the
      // walk's weight folded into the throughput
      PathState scaled = st;
      scaled.throughput = st.throughput * ss_weight;
      scaled.ray_d = ss_isect.w_i;
      store_camera_vertex(p, vertex_slot, scene, scaled, make_float4(ss_isect.bc.y, ss_isect.bc.z, ss_isect.t, __uint_as_float(ss_isect.tri)), derived.seed, &ss_isect, true,
        cb_vertex ? kCvNoConnect : 0u, opt_merge_vertices(it));
    } else {
      store_camera_vertex_m(p, vertex_slot, scene, st, h, derived.seed, &isect, step_material, false, 0u, opt_merge_vertices(it));
    }
  }
  // next event estimation; vcm_shared.hxx:1036-1046: after a walk, from the exit point scaled by the walk
  const bool from_exit = kWalk && subsurface_sampled;
  {  // general BSDFs: k_connect_endpoints (pipeline.h EndpointQueue)
    nee = nee && opt_connect_to_light(it) && (st.depth + 1 <= scene.max_path_length) && (st.depth + 1 >= scene.min_path_length);  // the early-outs of vcm_connect_to_light
    const uint32_t endpoint_slot = slots.get(nee, p.counters + kCntEndpoints);
    if (nee) {
      st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
      const Isect& from = from_exit ? ss_isect : isect;
      const float4 where = at_medium ? mk4(ms.pos, __uint_as_float(kInvalid)) : make_float4(from.bc.y, from.bc.z, from.t, __uint_as_float(from.tri));
      write_endpoint(p, endpoint_slot, where, at_medium ? st.ray_d : from.w_i, from_exit ? st.throughput * ss_weight : st.throughput, from_exit, st.d_vcm, st.d_vc, st.depth, st.medium, st.id,
        st.wavelength, st.sampler);
      st.sampler.pop_fixed();
    }
  }

  // ---- phase C
  return continue_path();
}

}  // namespace etxd
