// dev_vcm_steps.h - one segment of a light / camera sub path after the closest-hit query, shared by the wavefront shade
// kernels (kernels_vcm.hip: one launch per bounce) and the tail kernels (kernels_tail.hip: the few long paths that
// survive many bounces loop inside one launch).
#pragma once

#include "dev_vcm.h"

namespace etxd {

ETX_DEV void store_light_vertex(const Pipeline& p, const VcmParams& it, const PathState& st, const f3& pos, const f3& nrm, float bc_u, float bc_v, uint32_t tri, bool keep_bbox) {
  uint32_t idx = atomicAdd(p.counters + kCntLightVertices, 1u);
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
  const uint32_t prev = p.light_path_head[st.id];
  const uint32_t index_in_path = (prev == kInvalid) ? 0u : ((__float_as_uint(p.lv.bc_len_med(prev).z) >> 16u) + 1u);
  p.lv.bc_len_med(idx) = make_float4(bc_u, bc_v, __uint_as_float((index_in_path << 16u) | (st.depth & 0xffffu)), __uint_as_float(st.medium));
  p.lv.next(idx) = prev;
  p.light_path_head[st.id] = idx;
  (void)keep_bbox;  // the photon bounding box is reduced by k_grid_bbox (kernels_grid.hip)
}


ETX_DEV void store_camera_vertex(const Pipeline& p, const DScene& scene, const PathState& st, const float4& hit_or_pos, uint32_t seed, const Isect* isect) {
  uint32_t idx = atomicAdd(p.counters + kCntCameraVertices, 1u);
  if (idx >= p.capacity) {  // only the tail kernel can exceed one vertex per path slot
    atomicOr(p.counters + kCntOverflow, kOverflowCameraVertices);
    return;
  }
  p.cv.hit[idx] = hit_or_pos;
  p.cv.wi_medium[idx] = mk4(st.ray_d, __uint_as_float(st.medium));
  p.cv.thr_depth[idx] = mk4(st.throughput, __uint_as_float(st.depth));
  p.cv.mis_pixel[idx] = make_float4(st.d_vcm, st.d_vc, st.d_vm, __uint_as_float(st.id));
  p.cv.seed[idx] = seed;
  if (isect == nullptr) {
    p.cv.pos_info[idx] = make_float4(hit_or_pos.x, hit_or_pos.y, hit_or_pos.z, __uint_as_float((st.depth << 8u) | kCvMedium));
    return;
  }
  const etx_abi_material& mat = scene.materials[isect->material];
  const bool diffuse = mat.cls == ETX_MAT_DIFFUSE;
  f3 fthr = st.throughput;
  if (diffuse)
    fthr = fthr * apply_image(scene, mat.scattering, isect->tex, nullptr) * kInvPi;  // DiffuseBSDF func (bsdf_various.hxx:60-64) x t_camera
  p.cv.pos_info[idx] = mk4(isect->pos, __uint_as_float((st.depth << 8u) | (diffuse ? kCvDiffuse : 0u)));
  p.cv.nrm_dvm[idx] = mk4(isect->nrm, st.d_vm);
  p.cv.fthr_dvcm[idx] = mk4(fthr, st.d_vcm);
}


// vcm_light_step, vcm_shared.hxx:1090-1260: everything after rt.trace for one light sub path segment.
// Returns whether the path continues (state updated in place).
ETX_DEV bool light_step(const Pipeline& p, const DScene& scene, const VcmParams& it, PathState& st, const float4& h) {
  bool alive = false;
  const uint32_t tri = __float_as_uint(h.w);
  const bool found = tri != kInvalid;
  Isect isect;
  if (found)
    isect = make_intersection(scene, st.ray_d, h.x, h.y, h.z, tri);

  // vcm_try_sampling_medium, vcm_shared.hxx:379-388
  MediumSample ms;
  ms.sampled_medium_t = 0.0f;
  if (st.medium != kInvalid) {
    ms = sample_medium_homogeneous(scene.mediums[st.medium], st.throughput, st.sampler, st.ray_o, st.ray_d, found ? h.z : kMaxFloat);
    st.throughput *= ms.weight;
  }

  if (ms.sampled_medium()) {  // vcm_shared.hxx:1097-1170
    f2 rnd_bsdf = st.sampler.next_2d();
    f2 rnd_connection = st.sampler.next_2d();
    f2 rnd_support = st.sampler.next_2d();
    float seg = st.path_distance + ms.sampled_medium_t;
    st.d_vcm *= sqr(seg);
    st.path_distance = 0.0f;
    const DMedium& med = scene.mediums[st.medium];
    if (opt_connect_vertices(it) && (st.depth + 1 <= scene.max_path_length))
      store_light_vertex(p, it, st, ms.pos, mk3(0.0f), 0.0f, 0.0f, kInvalid, false);
    if (opt_connect_to_camera(it) && med.explicit_connections && (st.depth + 1 <= scene.max_path_length)) {
      st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
      vcm_connect_to_camera(p, scene, it, true, nullptr, ms.pos, st);
      st.sampler.pop_fixed();
    }
    f3 w_i = st.ray_d;
    f3 w_o = sample_phase_function(w_i, med.g, rnd_bsdf);
    float pdf_fwd = phase_function(w_i, w_o, med.g);
    float pdf_rev = phase_function(w_o, w_i, med.g);
    st.d_vc = (1.0f / pdf_fwd) * (st.d_vc * pdf_rev + st.d_vcm);
    st.d_vm = (1.0f / pdf_fwd) * (st.d_vm * pdf_rev + 0.0f);
    st.d_vcm = 1.0f / pdf_fwd;
    st.ray_o = ms.pos;
    st.ray_d = w_o;
    st.ray_tmax = kMaxFloat;
    st.ray_tmin = kRayEpsilon;
    st.depth += 1u;
    alive = (st.depth + 1 <= scene.max_path_length) && random_continue(st.depth, scene.random_path_termination, st.eta, st.sampler, st.throughput);
  } else if (found) {
    if (vcm_handle_boundary(scene, isect, st)) {
      alive = true;
    } else {
      const etx_abi_material& mat = scene.materials[isect.material];
      BsdfData bsdf_data = make_bsdf_data(isect, isect.w_i, st.medium, kPathLight);
      f2 rnd_bsdf = st.sampler.next_2d();
      f2 rnd_connection = st.sampler.next_2d();
      f2 rnd_support = st.sampler.next_2d();
      st.sampler.push_fixed(rnd_bsdf.x, rnd_bsdf.y, rnd_support.x);
      BsdfSample bs = bsdf_sample(scene, bsdf_data, mat, st.sampler);
      bool is_connectible = (bs.properties & kSampleDelta) == 0u;
      st.sampler.pop_fixed();

      // vcm_update_light_vcm, vcm_shared.hxx:451-461
      if ((st.depth > 0u) || (st.flags & kPathLocalEmitter))
        st.d_vcm *= sqr(st.path_distance + isect.t);
      float cos_to_prev = fabsf(dot(isect.nrm, -st.ray_d));
      st.d_vcm /= cos_to_prev;
      st.d_vc /= cos_to_prev;
      st.d_vm /= cos_to_prev;
      st.path_distance = 0.0f;

      if (is_connectible) {
        store_light_vertex(p, it, st, isect.pos, isect.nrm, isect.bc.y, isect.bc.z, isect.tri, true);
        if (opt_connect_to_camera(it) && (st.depth + 1 <= scene.max_path_length)) {
          st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
          vcm_connect_to_camera(p, scene, it, false, &isect, mk3(0.0f), st);
          st.sampler.pop_fixed();
        }
      }
      if (vcm_next_ray(scene, kPathLight, st, it, isect, bsdf_data, bs))
        alive = st.depth + 1u < scene.max_path_length;
    }
  }
  return alive;
}

// vcm_camera_step, vcm_shared.hxx:927-1079 after rt.trace, without the vertex connections and the merge: connectible
// vertices go to the camera vertex pool (k_expand_pairs / k_connect_pairs / k_merge consume them), NEE segments and the
// direct / miss radiance go to the shadow queue / film.
ETX_DEV bool camera_step(const Pipeline& p, const DScene& scene, const VcmParams& it, PathState& st, const float4& h) {
  bool alive = false;
  const uint32_t tri = __float_as_uint(h.w);
  const bool found = tri != kInvalid;
  f3 gathered = mk3(0.0f);
  Isect isect;
  if (found)
    isect = make_intersection(scene, st.ray_d, h.x, h.y, h.z, tri);

  MediumSample ms;
  ms.sampled_medium_t = 0.0f;
  if (st.medium != kInvalid) {
    ms = sample_medium_homogeneous(scene.mediums[st.medium], st.throughput, st.sampler, st.ray_o, st.ray_d, found ? h.z : kMaxFloat);
    st.throughput *= ms.weight;
  }

  if (ms.sampled_medium()) {  // vcm_shared.hxx:934-995
    f2 rnd_bsdf = st.sampler.next_2d();
    f2 rnd_connection = st.sampler.next_2d();
    f2 rnd_support = st.sampler.next_2d();
    float seg = st.path_distance + ms.sampled_medium_t;
    st.d_vcm *= sqr(seg);
    st.path_distance = 0.0f;
    const DMedium& med = scene.mediums[st.medium];
    f3 w_o_smp = sample_phase_function(st.ray_d, med.g, rnd_bsdf);
    float pdf_fwd = phase_function(st.ray_d, w_o_smp, med.g);
    float pdf_rev = phase_function(w_o_smp, st.ray_d, med.g);
    if (med.explicit_connections && (st.depth + 1 <= scene.max_path_length)) {
      if (opt_connect_to_light(it)) {
        st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
        vcm_connect_to_light(p, scene, it, true, nullptr, ms.pos, st, film_index(it, st.id));
        st.sampler.pop_fixed();
      }
      if (opt_connect_vertices(it)) {
        Sampler derived;
        derived.init(st.sampler.seed, 0x51ed270bu);
        store_camera_vertex(p, scene, st, mk4(ms.pos, __uint_as_float(kInvalid)), derived.seed, nullptr);
      }
    }
    st.d_vc = (1.0f / pdf_fwd) * (st.d_vc * pdf_rev + st.d_vcm);
    st.d_vm = (1.0f / pdf_fwd) * (st.d_vm * pdf_rev + 0.0f);
    st.d_vcm = 1.0f / pdf_fwd;
    st.ray_o = ms.pos;
    st.ray_d = w_o_smp;
    st.ray_tmax = kMaxFloat;
    st.ray_tmin = kRayEpsilon;
    st.depth += 1u;
    alive = (st.depth + 1 <= scene.max_path_length) && random_continue(st.depth, scene.random_path_termination, st.eta, st.sampler, st.throughput);
  } else if (found == false) {
    gathered += vcm_cam_handle_miss(scene, it, st);
  } else if (vcm_handle_boundary(scene, isect, st)) {
    alive = true;
  } else {
    const etx_abi_material& mat = scene.materials[isect.material];
    BsdfData bsdf_data = make_bsdf_data(isect, isect.w_i, st.medium, kPathCamera);
    f2 rnd_bsdf = st.sampler.next_2d();
    f2 rnd_connection = st.sampler.next_2d();
    f2 rnd_support = st.sampler.next_2d();
    // blue-noise override of the first vertex (vcm_shared.hxx:1018-1022) needs the host's tables:
    // etx_hip_begin rejects options.blue_noise until etx_hip_upload_bluenoise provided them.
    st.sampler.push_fixed(rnd_bsdf.x, rnd_bsdf.y, rnd_support.x);
    BsdfSample bs = bsdf_sample(scene, bsdf_data, mat, st.sampler);
    bool is_connectible = (bs.properties & kSampleDelta) == 0u;
    st.sampler.pop_fixed();

    // vcm_update_camera_vcm, vcm_shared.hxx:589-595
    float cos_to_prev = fabsf(dot(isect.nrm, -st.ray_d));
    st.d_vcm *= sqr(st.path_distance + isect.t) / cos_to_prev;
    st.d_vc /= cos_to_prev;
    st.d_vm /= cos_to_prev;
    st.path_distance = 0.0f;

    // vcm_handle_direct_hit, vcm_shared.hxx:597-606
    if (opt_direct_hit(it) && (isect.emitter != kInvalid) && (st.depth <= scene.max_path_length) && (st.depth >= scene.min_path_length))
      gathered += vcm_get_radiance(scene, scene.emitters[isect.emitter], st, it, isect);

    if (is_connectible) {
      if (opt_connect_vertices(it) || (opt_merge_vertices(it) && (st.depth + 1 <= scene.max_path_length))) {
        Sampler derived;
        derived.init(st.sampler.seed, 0x51ed270bu);
        store_camera_vertex(p, scene, st, h, derived.seed, &isect);
      }
      st.sampler.push_fixed(rnd_connection.x, rnd_connection.y, rnd_support.y);
      vcm_connect_to_light(p, scene, it, false, &isect, mk3(0.0f), st, film_index(it, st.id));
      st.sampler.pop_fixed();
    }
    alive = vcm_next_ray(scene, kPathCamera, st, it, isect, bsdf_data, bs);
  }
  if ((gathered.x != 0.0f) || (gathered.y != 0.0f) || (gathered.z != 0.0f))
    atomic_add_f3(p.camera_sum + film_index(it, st.id), gathered);
  return alive;
}

}  // namespace etxd
