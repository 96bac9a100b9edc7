// kernels_pt.hip - unidirectional path tracer (BASELINE configs[0]) on the same wavefront pipeline as VCM.
//
// Restates CPUPathTracingImpl::execute_range (sources/etx/rt/integrators/path_tracing.cxx:50-83) and
// run_path_iteration / handle_hit_ray / handle_sampled_medium / handle_missed_ray
// (sources/etx/rt/shared/path_tracing_shared.hxx:238-510):
//   k_pt_generate   make_ray_payload (:238-259): one payload per pixel, pixel filter + lens sample
//   k_trace_closest (kernels_trace.hip) = rt.trace
//   k_pt_shade      everything after rt.trace; the two transmittance queries of a segment (direct emitter hit :343-349,
//                   next event estimation :287-312 / :274-283) become shadow-queue requests against the pixel
//   k_trace_shadow  (kernels_trace.hip) multiplies by the transmittance and adds to the iteration image
//   k_pt_commit     path_tracing.cxx:67-82: radiance clamp of the iteration's pixel value (paths longer than one
//                   segment), accumulation into the camera image; normal / albedo AOVs are added by k_pt_shade
// PathState reuse: depth = path_length, d_vcm = sampled_bsdf_pdf, flags bit kPtMisWeight = payload.mis_weight.
// Not implemented (etx_hip_upload_scene / etx_hip_begin reject them): subsurface scattering, spectral mode.
#include "kernels_shade.inl"  // shading groups, bin_foreign_groups

namespace etxd {

static uint32_t grid_for(uint32_t capacity) {
  return min(kPersistentBlocks, (capacity + kBlockSize - 1) / kBlockSize);
}

enum : uint32_t { kPtMisWeight = 1u << 8 };

// bsdf::albedo, scene_bsdf.hxx:95-107 + bsdf_various.hxx:28,127,214,259,291 + bsdf_conductor.hxx:133
ETX_DEV f3 bsdf_albedo(const DScene& scene, const etx_abi_material& mat, const f2 tex, float wavelength) {
  switch (mat.cls) {
    case ETX_MAT_DIFFUSE:
    case ETX_MAT_TRANSLUCENT:
    case ETX_MAT_PLASTIC:     // bsdf_plastic.hxx:182
    case ETX_MAT_DIELECTRIC:  // bsdf_dielectric.hxx:256
    case ETX_MAT_THINFILM:    // bsdf_dielectric.hxx:55
    case ETX_MAT_VELVET:      // bsdf_velvet.hxx:122
    case ETX_MAT_PRINCIPLED:  // bsdf_principled.hxx:120
      return apply_image(scene, mat.scattering, tex, nullptr, wavelength);
    case ETX_MAT_CONDUCTOR:
      return apply_image(scene, mat.reflectance, tex, nullptr, wavelength);
    case ETX_MAT_MIRROR:
    case ETX_MAT_BOUNDARY:
      return mk3(1.0f);
    default:
      return mk3(0.0f);
  }
}

// Film::sample, film.cxx:137-145
ETX_DEV f2 film_sample(const DScene& scene, bool filtered, uint32_t px, uint32_t py, const VcmParams& it, const f2 rnd) {
  f2 jitter = {rnd.x * 2.0f - 1.0f, rnd.y * 2.0f - 1.0f};
  float radius = 0.0f;  // PixelFilter::empty() for the first iteration (path_tracing_shared.hxx:245)
  if (filtered) {
    radius = scene.pixel_sampler_radius;
    if (scene.pixel_sampler_image != kInvalid) {
      float pdf = 0.0f;
      float4 eval;
      f2 uv = image_sample(scene.images[scene.pixel_sampler_image], rnd, pdf, eval);
      jitter = {uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f};
    }
  }
  return {(float(px) + 0.5f + radius * jitter.x) / float(it.film_w) * 2.0f - 1.0f, (float(py) + 0.5f + radius * jitter.y) / float(it.film_h) * 2.0f - 1.0f};
}

// make_ray_payload, path_tracing_shared.hxx:238-259 (RGB mode)
__global__ __launch_bounds__(kBlockSize) void k_pt_generate(Pipeline p, VcmParams it) {
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  // Film::active_pixel (film.cxx:434-459): converged pixels are not sampled; the others are marked "sampled" in the iteration
  // image (k_pt_commit adds nothing to pixels that were not)
  ETX_BLOCK_LOOP(it.path_count, k) {
    const bool in_range = k < it.path_count;
    const uint32_t id = path_pixel(it, k);
    const uint32_t storage = in_range ? film_index(it, id) : 0u;
    const bool active = in_range && ((p.pixel_state == nullptr) || ((p.pixel_state[storage] & 1u) == 0u));
    const uint32_t slot = block_compact_slot(active, p.counters + kCntActiveA, s_scratch);
    const uint32_t block_active = __syncthreads_count(active);
    if ((threadIdx.x == 0u) && (block_active != 0u))
      atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatActivePixels), (unsigned long long)block_active);
    if (active == false)
      continue;
    p.camera_sum[storage].w = 1.0f;
    PathState st;
    st.id = id;
    st.sampler.init(id, it.iteration);
    st.wavelength = scene.spectral ? spectral_sample_wavelength(st.sampler.next()) : 0.0f;  // path_tracing_shared.hxx:243
    const uint32_t px = id % it.film_w, py = id / it.film_w;
    const f2 uv = film_sample(scene, it.iteration != 0u, px, py, it, st.sampler.next_2d());
    RayGen r = generate_ray(scene, uv, st.sampler.next_2d());
    st.ray_o = r.o, st.ray_d = r.d, st.ray_tmin = r.tmin, st.ray_tmax = r.tmax;
    st.throughput = mk3(1.0f);
    st.medium = scene.camera.medium_index;
    st.depth = 1u;
    st.eta = 1.0f;
    st.d_vcm = 0.0f;  // sampled_bsdf_pdf
    st.d_vc = st.d_vm = st.path_distance = 0.0f;
    st.flags = kPtMisWeight;
    store_path(p.paths[0], slot, st);
  }
}

struct PtRequests {
  ShadowRequest direct, nee;
  bool has_direct, has_nee;
};

// run_path_iteration after rt.trace, path_tracing_shared.hxx:479-508
template <uint32_t kGroup>
ETX_DEV bool pt_step(const Pipeline& p, const DScene& scene, const VcmParams& it, PathState& st, const float4& h, PtRequests& out, const LaneStack& stack) {
  constexpr bool kSimple = kGroup == kShadeGroupSimple;
  constexpr bool kWalk = kGroup == kShadeGroupSubsurface;
  const bool opt_direct = (it.options & ETX_PT_DIRECT) != 0u, opt_nee = (it.options & ETX_PT_NEE) != 0u, opt_mis = (it.options & ETX_PT_MIS) != 0u;
  if (st.depth > scene.max_path_length)
    return false;
  const uint32_t film_target = film_index(it, st.id);
  const f3 film_weight = spectral_film_weight(scene, st.wavelength);  // (value / sampling_pdf).to_rgb(), path_tracing.cxx:67-72
  const uint32_t tri_index = __float_as_uint(h.w);
  const bool found = tri_index != kInvalid;
  Isect isect;
  if (found)
    isect = make_intersection(scene, st.ray_d, h.x, h.y, h.z, tri_index);

  // try_sampling_medium, :261-270
  MediumSample ms;
  ms.sampled_medium_t = 0.0f;
  if (st.medium != kInvalid) {
    ms = sample_medium_homogeneous(scene, scene.mediums[st.medium], st.wavelength, st.throughput, st.sampler, st.ray_o, st.ray_d, found ? h.z : kMaxFloat);
    st.throughput *= ms.weight;
  }

  if (ms.sampled_medium()) {  // handle_sampled_medium, :272-298
    const DMedium& medium = scene.mediums[st.medium];
    if (opt_nee && (st.depth + 1u <= scene.max_path_length) && medium.explicit_connections) {
      const uint32_t emitter_index = sample_emitter_index(scene, st.sampler.next());
      const EmitterSample es = sample_emitter(scene, emitter_index, st.sampler.next_2d(), ms.pos, st.wavelength);
      if (es.pdf_dir > 0.0f) {
        const float phase = phase_function(st.ray_d, es.direction, medium.g);
        const float weight = es.is_delta ? 1.0f : power_heuristic(es.pdf_dir * es.pdf_sample, phase);
        out.nee = {ms.pos, es.origin, st.throughput * es.value * (phase * weight / (es.pdf_dir * es.pdf_sample)) * film_weight, st.medium, film_target, st.wavelength};
        out.has_nee = true;
      }
    }
    const f3 w_o = sample_phase_function(st.ray_d, medium.g, st.sampler.next_2d());
    st.d_vcm = phase_function(st.ray_d, w_o, medium.g);
    st.flags |= kPtMisWeight;
    st.ray_o = ms.pos;
    st.ray_d = w_o;
    st.ray_tmax = kMaxFloat;
    st.ray_tmin = kRayEpsilon;
    st.depth += 1u;
    p.camera_sum[film_target].w = 2.0f;  // the path is longer than one segment (radiance clamp, path_tracing.cxx:74)
    return random_continue(st.depth, scene.random_path_termination, st.eta, st.sampler, st.throughput);
  }

  if (found == false) {  // handle_missed_ray, :458-477
    if (opt_direct) {
      f3 accumulated = mk3(0.0f);
      for (uint32_t ie = 0; ie < scene.env_count; ++ie) {
        const etx_abi_emitter& em = scene.emitters[scene.env_emitters[ie]];
        EmitterRadianceQuery q;
        q.source_position = q.target_position = mk3(0.0f);
        q.direction = st.ray_d;
        q.uv = {0.0f, 0.0f};
        q.directly_visible = st.depth == 1u;
        float pdf_area = 0.0f, pdf_dir = 0.0f, pdf_dir_out = 0.0f;
        const f3 e = emitter_get_radiance(scene, em, q, pdf_area, pdf_dir, pdf_dir_out, st.wavelength);
        if ((pdf_dir > 0.0f) && (is_zero(e) == false)) {
          const float pdf_discrete = emitter_discrete_pdf(scene, em);
          const float weight = (((st.flags & kPtMisWeight) == 0u) || q.directly_visible) ? 1.0f : power_heuristic(st.d_vcm, pdf_discrete * pdf_dir);
          accumulated += st.throughput * e * weight;
        }
      }
      if ((accumulated.x != 0.0f) || (accumulated.y != 0.0f) || (accumulated.z != 0.0f))
        film_add(p, p.camera_sum + film_target, accumulated * film_weight);
    }
    return false;
  }

  // handle_hit_ray, :352-456
  const etx_abi_triangle& tri = scene.triangles[isect.tri];
  const etx_abi_material& mat = scene.materials[isect.material];
  if (mat.cls == ETX_MAT_BOUNDARY) {
    st.medium = (dot(isect.nrm, st.ray_d) < 0.0f) ? mat.int_medium : mat.ext_medium;
    st.ray_o = shading_pos(scene, tri, isect.bc, st.ray_d);
    st.ray_tmax = kMaxFloat;
    st.ray_tmin = kRayEpsilon;
    return true;
  }

  // handle_direct_emitter, :323-350
  if (opt_direct && (isect.emitter != kInvalid)) {
    const etx_abi_emitter& em = scene.emitters[isect.emitter];
    EmitterRadianceQuery q;
    q.source_position = st.ray_o;
    q.target_position = isect.pos;
    q.direction = mk3(0.0f);
    q.uv = isect.tex;
    q.directly_visible = st.depth == 1u;
    float pdf_area = 0.0f, pdf_dir = 0.0f, pdf_dir_out = 0.0f;
    const f3 e = emitter_get_radiance(scene, em, q, pdf_area, pdf_dir, pdf_dir_out, st.wavelength);
    if (pdf_dir > 0.0f) {
      const float pdf_discrete = emitter_discrete_pdf(scene, em);
      const bool no_weight = (opt_mis == false) || q.directly_visible || ((st.flags & kPtMisWeight) == 0u);
      const float weight = no_weight ? 1.0f : power_heuristic(st.d_vcm, pdf_discrete * pdf_dir);
      out.direct = {st.ray_o, isect.pos, st.throughput * e * weight * film_weight, st.medium, film_target, st.wavelength};
      out.has_direct = true;
    }
  }

  BsdfData bsdf_data = make_bsdf_data(isect, isect.w_i, st.medium, kPathCamera, st.wavelength);
  if (st.depth == 1u) {  // view_normal / view_albedo -> Film::accumulate_camera_image(pixel, color, normal, albedo)
    const f3 albedo = bsdf_albedo(scene, mat, isect.tex, st.wavelength) * film_weight;
    film_add(p, p.normal_sum + film_target, isect.nrm);  // atomics: another lane may add the same pixel of another iteration
    film_add(p, p.albedo_sum + film_target, albedo);
  }

  f2 rnd_bsdf = st.sampler.next_2d();
  f2 rnd_em_sample = st.sampler.next_2d();
  f2 rnd_support = st.sampler.next_2d();
  if ((it.bluenoise != nullptr) && (st.depth == 1u))  // :380-384 (no iteration limit here, the sampler wraps at 256)
    bluenoise_samples(it.bluenoise, st.id % it.film_w, st.id / it.film_w, it.iteration, rnd_bsdf, rnd_em_sample, rnd_support);

  st.sampler.push_fixed(rnd_bsdf.x, rnd_bsdf.y, rnd_support.x);
  const BsdfSample bs = bsdf_sample_s<kSimple>(scene, bsdf_data, mat, st.sampler);
  st.sampler.pop_fixed();
  // path_tracing_shared.hxx:391-408: a diffuse reflection off a subsurface material enters the object instead
  bool subsurface_sampled = false, gathered_light = false;
  Isect ss_isect;
  f3 ss_weight = mk3(0.0f);
  if (kWalk && (mat.subsurface.cls != 0u) && (bs.properties & kSampleReflection) && (bs.properties & kSampleDiffuse)) {
    if (mat.subsurface.cls == 2u) {
      // Christensen-Burley (subsurface::gather_cb): the light is gathered at EVERY exit point with its weight (:419-426); the
      // exit points are not buffered (dev_sss.h), so their next-event requests are written as they are found, one queue slot
      // per lane-level reservation. bsdf_sample.valid() and the medium change are decided before (:405-409).
      const bool light_them = opt_nee && (st.depth + 1u <= scene.max_path_length) && bs.valid();
      const uint32_t emitter_index = sample_emitter_index(scene, rnd_support.y);
      const uint32_t nee_medium = (bs.properties & kSampleMediumChanged) ? bs.medium_index : st.medium;
      const etx_abi_material& exit_mat = scene.materials[scene.subsurface_exit_material];
      gathered_light = true;
      subsurface_sampled = sss_gather_cb(scene, stack, isect, st.sampler, st.wavelength, ss_isect, ss_weight, [&](const Isect& exit_point, const f3& weight) {
        if (light_them == false)
          return;
        const EmitterSample es = sample_emitter(scene, emitter_index, rnd_em_sample, exit_point.pos, st.wavelength);
        if (es.pdf_dir == 0.0f)
          return;
        const BsdfData nee_data = make_bsdf_data(exit_point, exit_point.w_i, nee_medium, kPathCamera, st.wavelength);
        const BsdfEval eval = bsdf_evaluate_s<kSimple>(scene, nee_data, es.direction, exit_mat, st.sampler);
        if (eval.valid() == false)
          return;
        const f3 pos = shading_pos(scene, scene.triangles[exit_point.tri], exit_point.bc, es.direction);
        const float mis = ((opt_mis == false) || es.is_delta) ? 1.0f : power_heuristic(es.pdf_dir * es.pdf_sample, eval.pdf);
        const f3 value = st.throughput * weight * eval.bsdf * es.value * (mis / (es.pdf_dir * es.pdf_sample)) * film_weight;
        write_shadow(p, atomicAdd(p.counters + kCntShadow, 1u), ShadowRequest{pos, es.origin, value, nee_medium, film_target, st.wavelength});
      });
    } else {
      subsurface_sampled = sss_gather_rw(scene, stack, isect, st.sampler, st.wavelength, ss_isect, ss_weight);
    }
    if (subsurface_sampled == false)
      return false;
  }
  if (bs.valid() == false)
    return false;
  if (bs.properties & kSampleMediumChanged)
    st.medium = bs.medium_index;

  if (opt_nee && (gathered_light == false) && (st.depth + 1u <= scene.max_path_length)) {  // :409-431 + evaluate_light :300-321
    st.sampler.push_fixed(rnd_em_sample.x, rnd_em_sample.y, rnd_support.x);
    const uint32_t emitter_index = sample_emitter_index(scene, rnd_support.y);
    // :414-421: after a subsurface walk the light is gathered at the exit point through scene.subsurface_exit_material
    const Isect& nee_isect = subsurface_sampled ? ss_isect : isect;
    const etx_abi_material& nee_mat = subsurface_sampled ? scene.materials[scene.subsurface_exit_material] : mat;
    const f3 nee_scale = subsurface_sampled ? ss_weight : mk3(1.0f);
    const EmitterSample es = sample_emitter(scene, emitter_index, rnd_em_sample, nee_isect.pos, st.wavelength);
    if (es.pdf_dir != 0.0f) {
      BsdfData nee_data = make_bsdf_data(nee_isect, nee_isect.w_i, st.medium, kPathCamera, st.wavelength);
      const BsdfEval eval = bsdf_evaluate_s<kSimple>(scene, nee_data, es.direction, nee_mat, st.sampler);
      if (eval.valid()) {
        const f3 pos = shading_pos(scene, scene.triangles[nee_isect.tri], nee_isect.bc, es.direction);
        const bool no_weight = (opt_mis == false) || es.is_delta;
        const float weight = no_weight ? 1.0f : power_heuristic(es.pdf_dir * es.pdf_sample, eval.pdf);
        out.nee = {pos, es.origin, st.throughput * nee_scale * eval.bsdf * es.value * (weight / (es.pdf_dir * es.pdf_sample)) * film_weight, st.medium, film_target, st.wavelength};
        out.has_nee = true;
      }
    }
    st.sampler.pop_fixed();
  }

  if (subsurface_sampled) {  // :433-439
    st.ray_d = sample_cosine_distribution(rnd_bsdf, ss_isect.nrm, 1.0f);
    st.throughput *= ss_weight;  // weights[selected] * selected_sample_weight (= 1 for the random walk)
    st.d_vcm = fabsf(dot(st.ray_d, ss_isect.nrm)) / kPi;
    st.flags |= kPtMisWeight;
    st.ray_o = shading_pos(scene, scene.triangles[ss_isect.tri], ss_isect.bc, st.ray_d);
  } else {
    st.throughput *= bs.weight;
    st.d_vcm = bs.pdf;
    st.flags = bs.is_delta() ? (st.flags & ~kPtMisWeight) : (st.flags | kPtMisWeight);
    st.eta *= bs.eta;
    st.ray_d = bs.w_o;
    st.ray_o = shading_pos(scene, tri, isect.bc, st.ray_d);
  }
  if (is_zero(st.throughput))
    return false;
  st.ray_tmax = kMaxFloat;
  st.ray_tmin = kRayEpsilon;
  st.depth += 1u;
  p.camera_sum[film_target].w = 2.0f;
  return random_continue(st.depth, scene.random_path_termination, st.eta, st.sampler, st.throughput);
}

template <uint32_t kGroup, bool kBin>
__global__ __launch_bounds__(kBlockSize) void k_pt_shade(Pipeline p, VcmParams it, uint32_t in_set) {
  constexpr bool kWalk = kGroup == kShadeGroupSubsurface;
  __shared__ BlockScratch s_scratch;
  __shared__ int32_t s_stack[kWalk ? kStackDepth * kBlockSize : 1];  // the subsurface walk traverses inline
  const LaneStack stack = lane_stack(p.scene, s_stack + (kWalk ? threadIdx.x : 0u), kBlockSize);
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const PathSet& out = p.paths[in_set ^ 1u];
  const uint32_t count = shade_item_count<kGroup>(p, in_set);
  uint32_t* out_counter = p.counters + (in_set == 0 ? kCntActiveB : kCntActiveA);
  const BlockSlots slots = {&s_scratch, nullptr};
  ETX_BLOCK_LOOP(count, j) {
    bool valid = j < count;
    const uint32_t i = (kGroup == kShadeGroupSimple) ? j : (valid ? p.group_list[kGroup == kShadeGroupSimple ? 0u : kGroup - 1u][j] : 0u);
    PathState st;
    PtRequests requests;
    requests.has_direct = requests.has_nee = false;
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (valid)
      h = p.hits[i];
    if (kBin)
      valid = bin_foreign_groups(p, slots, i, hit_shade_group(scene, h), valid);
    bool alive = false;
    if (valid) {
      st = load_path(in, i);
      alive = pt_step<kGroup>(p, scene, it, st, h, requests, stack);
    }
    const uint32_t direct_slot = slots.get(requests.has_direct, p.counters + kCntShadow);
    if (requests.has_direct)
      write_shadow(p, direct_slot, requests.direct);
    const uint32_t nee_slot = slots.get(requests.has_nee, p.counters + kCntShadow);
    if (requests.has_nee)
      write_shadow(p, nee_slot, requests.nee);
    const uint32_t slot = slots.get(alive, out_counter);
    if (alive)
      store_path(out, slot, st);
  }
}

// path_tracing.cxx:67-82 + Film::accumulate_camera_image (film.cxx:173-231): clamp the iteration's pixel value, add it to
// the camera image. `iteration_image` holds the sum of the iteration's contributions in xyz; w: 0 = the pixel was not sampled
// (converged), 1 = sampled, 2 = sampled and the path continued past its first vertex (the clamp applies).
__global__ __launch_bounds__(kBlockSize) void k_pt_commit(float4* __restrict__ iteration_image, float4* __restrict__ camera_sum, float4* __restrict__ adaptive_sum, uint32_t pixels,
  float radiance_clamp) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += gridDim.x * blockDim.x) {
    const float4 v = iteration_image[i];
    if (v.w == 0.0f)
      continue;
    f3 color = {v.x, v.y, v.z};
    if ((radiance_clamp > 0.0f) && (v.w == 2.0f)) {
      const float lum = luminance(color);
      if (lum > radiance_clamp)
        color *= radiance_clamp / lum;
    }
    atomic_add_f3(camera_sum + i, color);  // the film is shared by the lanes (host_api.cpp)
    const float sample_index = atomicAdd(&camera_sum[i].w, 1.0f);  // samples committed to this pixel before this one
    if ((adaptive_sum != nullptr) && ((uint32_t(sample_index) & 1u) == 0u)) {  // film.cxx:218-225: the mean of the even samples
      atomic_add_f3(adaptive_sum + i, color);
      atomicAdd(&adaptive_sum[i].w, 1.0f);
    }
    iteration_image[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

// Film::estimate_noise_levels, film.cxx:233-330. Pass 1: error level of every pixel that is still sampled.
__global__ __launch_bounds__(kBlockSize) void k_noise_estimate(const float4* __restrict__ camera_sum, const float4* __restrict__ adaptive_sum, uint32_t* __restrict__ pixel_state,
  uint32_t pixels, float threshold) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += gridDim.x * blockDim.x) {
    if (pixel_state[i] & 1u)
      continue;
    const float4 c = camera_sum[i], a = adaptive_sum[i];
    const float inv_c = (c.w > 0.0f) ? 1.0f / c.w : 0.0f, inv_a = (a.w > 0.0f) ? 1.0f / a.w : 0.0f;
    const f3 v_i = f3{c.x, c.y, c.z} * inv_c, v_a = f3{a.x, a.y, a.z} * inv_a;
    const float error_diff = fabsf(v_i.x - v_a.x) + fabsf(v_i.y - v_a.y) + fabsf(v_i.z - v_a.z);
    const float error_norm = fabsf(v_i.x) + fabsf(v_i.y) + fabsf(v_i.z);
    const float error_level = error_diff / (((error_norm < 1.0f) ? sqrtf(error_norm) : error_norm) + kEpsilon);
    const uint32_t converged = (error_level < threshold) ? 1u : 0u;
    pixel_state[i] = converged | (converged << 1u);  // converged, tmp = converged
  }
}
// Pass 2: a pixel that is still sampled keeps its row neighbours [x - 5, x + 5) "not converged" in tmp (kBlockSize 5, :283-299)
__global__ __launch_bounds__(kBlockSize) void k_noise_spread_rows(uint32_t* __restrict__ pixel_state, uint32_t width, uint32_t pixels) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += gridDim.x * blockDim.x) {
    if (pixel_state[i] & 1u)
      continue;
    const uint32_t x = i % width, y = i / width;
    const uint32_t begin_x = (x >= 5u) ? x - 5u : 0u, end_x = min(width, x + 5u);
    for (uint32_t q = begin_x; q < end_x; ++q)
      atomicAnd(pixel_state + q + y * width, ~2u);
  }
}
// Pass 3: every pixel whose tmp is clear re-activates its column neighbours [y - 5, y + 5) (:305-321)
__global__ __launch_bounds__(kBlockSize) void k_noise_spread_columns(uint32_t* __restrict__ pixel_state, uint32_t width, uint32_t height, uint32_t pixels) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += gridDim.x * blockDim.x) {
    if (pixel_state[i] & 2u)
      continue;
    const uint32_t x = i % width, y = i / width;
    const uint32_t begin_y = (y >= 5u) ? y - 5u : 0u, end_y = min(height, y + 5u);
    for (uint32_t q = begin_y; q < end_y; ++q)
      atomicAnd(pixel_state + x + q * width, ~1u);
  }
}

void launch_pt_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it) {
  hipLaunchKernelGGL(k_pt_generate, dim3(grid_for(p.capacity)), dim3(kBlockSize), 0, stream, p, it);
}

void launch_pt_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups) {
  const dim3 grid(max(1u, grid_for(min(p.capacity, max_items))));
  if (groups.binned() == false) {
    hipLaunchKernelGGL((k_pt_shade<kShadeGroupSimple, false>), grid, dim3(kBlockSize), 0, stream, p, it, in_set);
    return;
  }
  hipLaunchKernelGGL((k_pt_shade<kShadeGroupSimple, true>), grid, dim3(kBlockSize), 0, stream, p, it, in_set);
  if (groups.general)
    hipLaunchKernelGGL((k_pt_shade<kShadeGroupGeneral, false>), grid, dim3(kBlockSize), 0, stream, p, it, in_set);
  if (groups.subsurface)
    hipLaunchKernelGGL((k_pt_shade<kShadeGroupSubsurface, false>), grid, dim3(kBlockSize), 0, stream, p, it, in_set);
}

void launch_noise_estimate(hipStream_t stream, const Pipeline& p, uint32_t width, uint32_t height, float threshold) {
  const uint32_t pixels = width * height;
  hipLaunchKernelGGL(k_noise_estimate, dim3(grid_for(pixels)), dim3(kBlockSize), 0, stream, p.camera_sum, p.adaptive_sum, p.pixel_state, pixels, threshold);
  hipLaunchKernelGGL(k_noise_spread_rows, dim3(grid_for(pixels)), dim3(kBlockSize), 0, stream, p.pixel_state, width, pixels);
  hipLaunchKernelGGL(k_noise_spread_columns, dim3(grid_for(pixels)), dim3(kBlockSize), 0, stream, p.pixel_state, width, height, pixels);
}

void launch_pt_commit(hipStream_t stream, float4* iteration_image, float4* camera_sum, float4* adaptive_sum, uint32_t pixels, float radiance_clamp) {
  hipLaunchKernelGGL(k_pt_commit, dim3(grid_for(pixels)), dim3(kBlockSize), 0, stream, iteration_image, camera_sum, adaptive_sum, pixels, radiance_clamp);
}

}  // namespace etxd
