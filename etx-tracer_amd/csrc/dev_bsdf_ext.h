// dev_bsdf_ext.h - the remaining BSDF classes of scene_bsdf.hxx:56-107 (RGB mode): Dielectric, Thinfilm, Plastic, Velvet,
// Principled and the rough-diffuse variations of DiffuseBSDF. Included by dev_bsdf.h between the microfacet helpers
// and the dispatch; every function cites the reference lines it restates.
#pragma once

namespace etxd {

// ---------------------------------------------------------------------------------------------------------------
// Heitz multiple-scattering microfacet model, dielectric part (bsdf_external.hxx:355-555)

ETX_DEV float ms_abgam(float x) {  // bsdf_external.hxx:355-359
  const float gam[7] = {1.0f / 12.0f, 1.0f / 30.0f, 53.0f / 210.0f, 195.0f / 371.0f, 22999.0f / 22737.0f, 29944523.0f / 19733142.0f, 109535241009.0f / 48264275462.0f};
  const float kHalfLogDoublePi = 0.918938518f;
  return kHalfLogDoublePi - x + (x - 0.5f) * logf(x) + gam[0] / (x + gam[1] / (x + gam[2] / (x + gam[3] / (x + gam[4] / (x + gam[5] / (x + gam[6] / x))))));
}
ETX_DEV float ms_gamma(float x) {  // :361-363
  return expf(ms_abgam(x + 5.0f)) / (x * (x + 1.0f) * (x + 2.0f) * (x + 3.0f) * (x + 4.0f));
}
ETX_DEV float ms_beta(float m, float n) {  // :365-367
  return ms_gamma(m) * ms_gamma(n) / ms_gamma(m + n);
}
ETX_DEV f3 ms_refract(const f3& wi, const f3& wm, float eta) {  // :369-374
  const float cos_theta_i = dot(wi, wm);
  const float cos_theta_t2 = 1.0f - (1.0f - cos_theta_i * cos_theta_i) / (eta * eta);
  const float cos_theta_t = -sqrtf(fmaxf(0.0f, cos_theta_t2));
  return wm * (dot(wi, wm) / eta + cos_theta_t) - wi / eta;
}

ETX_DEV float ior_eta_ratio(const Ior& num, const Ior& den) {  // (a.eta / b.eta).monochromatic(), spectrum.hxx:303-305
  return luminance(num.eta / den.eta);
}

// evalPhaseFunction_dielectric, :377-404 (by convention the ray is outside)
ETX_DEV f3 ms_eval_phase_dielectric(const MsRay& ray, const f3& wo, bool reflection, const Ior& ext_ior, const Ior& int_ior, const ThinfilmEval& tf, const f2 alpha) {
  if (ray.w.z > 0.9999f)
    return mk3(0.0f);
  if (reflection)
    return ms_phase_function_reflection(ray, wo, alpha, ext_ior, int_ior, tf);
  float projected_area = (ray.w.z < -0.9999f) ? 1.0f : ray.Lambda * ray.w.z;
  if (projected_area < kEpsilon)
    return mk3(0.0f);
  const float eta = ior_eta_ratio(int_ior, ext_ior);
  f3 wh = normalize(-ray.w + wo * eta);
  wh *= (wh.z > 0.0f) ? 1.0f : -1.0f;
  const float i_dot_m = -dot(wh, ray.w);
  if (i_dot_m < 0.0f)
    return mk3(0.0f);
  const float o_dot_m = dot(wo, wh);
  const float scalar = eta * eta * i_dot_m * fmaxf(0.0f, -o_dot_m) * D_ggx(wh, alpha) / (projected_area * sqr(i_dot_m + eta * o_dot_m));
  const f3 f = fresnel_calculate_g(i_dot_m, ext_ior, int_ior, tf);
  return (mk3(1.0f) - f) * scalar;
}

struct MsDielectricSample {  // :407-411
  f3 w_o, weight;
  bool reflection;
};

// samplePhaseFunction_dielectric, :413-451 (by convention wi is outside)
ETX_DEV MsDielectricSample ms_sample_phase_dielectric(const f2 rnd_slope, float rnd_reflection, const f3& wi, const f2 alpha, const Ior& ext_ior, const Ior& int_ior,
  const ThinfilmEval& tf) {
  const f3 wi_11 = normalize(f3{alpha.x * wi.x, alpha.y * wi.y, wi.z});
  f2 slope_11 = ms_sample_p22_11(acosf(wi_11.z), rnd_slope);
  const float phi = atan2f(wi_11.y, wi_11.x);
  float sp, cp;
  sincosf(phi, &sp, &cp);
  f2 slope = {cp * slope_11.x - sp * slope_11.y, sp * slope_11.x + cp * slope_11.y};
  slope.x *= alpha.x;
  slope.y *= alpha.y;
  f3 wm;
  if ((slope.x != slope.x) || isinf(slope.x))
    wm = (wi.z > 0.0f) ? f3{0.0f, 0.0f, 1.0f} : normalize(f3{wi.x, wi.y, 0.0f});
  else
    wm = normalize(f3{-slope.x, -slope.y, 1.0f});
  const float i_dot_m = dot(wi, wm);
  const f3 f = fresnel_calculate_g(i_dot_m, ext_ior, int_ior, tf);
  const float eta = ior_eta_ratio(int_ior, ext_ior);
  MsDielectricSample r;
  r.reflection = rnd_reflection < luminance(f);
  r.weight = r.reflection ? f : (mk3(1.0f) - f);
  r.w_o = r.reflection ? (-wi + 2.0f * wm * i_dot_m) : normalize(ms_refract(wi, wm, eta));
  return r;
}

ETX_DEV float ms_mis_weight_dielectric(const f3& wi, const f3& wo, bool reflection, float eta, const f2 alpha) {  // :454-464
  if (reflection) {
    if (wi.x == -wo.x && wi.y == -wo.y && wi.z == -wo.z)
      return 1.0f;
    const f3 wh = normalize(wi + wo);
    return D_ggx((wh.z > 0.0f) ? wh : -wh, alpha);
  }
  const f3 wh = normalize(wi + wo * eta);
  return D_ggx((wh.z > 0.0f) ? wh : -wh, alpha);
}

// eval_dielectric, :466-555 (stochastic)
ETX_DEV f3 ms_eval_dielectric_inline(Sampler& smp, const f3& wi, const f3& wo, bool wo_outside, const f2 alpha, const Ior& ext_ior, const Ior& int_ior, const ThinfilmEval& tf) {
  if ((wi.z <= 0.0f) || (wo.z <= 0.0f && wo_outside) || (wo.z >= 0.0f && !wo_outside))
    return mk3(0.0f);
  MsRay ray = ms_ray(-wi, alpha);
  ray.update_height(1.0f);
  bool outside = true;
  MsRay ray_shadowing = ms_ray(wo_outside ? wo : -wo, alpha);
  f3 single_scattering = mk3(0.0f), multiple_scattering = mk3(0.0f);
  float wi_mis_weight = 0.0f;
  const float eta = ior_eta_ratio(int_ior, ext_ior);
  uint32_t order = 0;
  while (order < kScatteringOrderMax) {
    ray.update_height(ms_sample_height(ray, smp.next()));
    if (ray.h == kMaxFloat)
      break;
    order++;
    if (order == 1) {
      const f3 phase = ms_eval_phase_dielectric(ray, wo, wo_outside, ext_ior, int_ior, tf, alpha);
      float G2_G1;
      if (wo_outside)
        G2_G1 = (1.0f + (-ray.Lambda - 1.0f)) / (1.0f + (-ray.Lambda - 1.0f) + ray_shadowing.Lambda);
      else
        G2_G1 = (1.0f + (-ray.Lambda - 1.0f)) * ms_beta(1.0f + (-ray.Lambda - 1.0f), 1.0f + ray_shadowing.Lambda);
      if (isfinite(G2_G1))
        single_scattering = phase * G2_G1;
    }
    if (order > 1) {
      f3 phase;
      float mis;
      if (outside) {
        phase = ms_eval_phase_dielectric(ray, wo, wo_outside, ext_ior, int_ior, tf, alpha);
        mis = wi_mis_weight / (wi_mis_weight + ms_mis_weight_dielectric(-ray.w, wo, wo_outside, eta, alpha));
      } else {
        phase = ms_eval_phase_dielectric(ray, -wo, !wo_outside, int_ior, ext_ior, tf, alpha);
        mis = wi_mis_weight / (wi_mis_weight + ms_mis_weight_dielectric(-ray.w, -wo, !wo_outside, 1.0f / eta, alpha));
      }
      ray_shadowing.update_height((outside == wo_outside) ? ray.h : -ray.h);
      multiple_scattering += phase * (ray_shadowing.G1 * mis);
    }
    const f2 rnd_slope = ((order == 1) && smp.has_fixed()) ? f2{smp.fixed_u, smp.fixed_v} : smp.next_2d();
    const float rnd_reflection = ((order == 1) && smp.has_fixed()) ? smp.fixed_w : smp.next();
    const MsDielectricSample next = ms_sample_phase_dielectric(rnd_slope, rnd_reflection, -ray.w, alpha, outside ? ext_ior : int_ior, outside ? int_ior : ext_ior, tf);
    if (next.reflection) {
      ray.update_direction(next.w_o, alpha);
      ray.update_height(ray.h);
    } else {
      outside = !outside;
      ray.update_direction(-next.w_o, alpha);
      ray.update_height(-ray.h);
    }
    if (order == 1)
      wi_mis_weight = ms_mis_weight_dielectric(wi, ray.w, outside, eta, alpha);
    if ((ray.h != ray.h) || (ray.w.x != ray.w.x) || (ray.w.z <= kEpsilon))
      return mk3(0.0f);
  }
  return 0.5f * single_scattering + multiple_scattering;
}

// one real function for the walk (DielectricBSDF::evaluate and the specular layer of PlasticBSDF share it)
struct MsEvalRet {
  f3 value;
  uint32_t seed;
};
static __device__ __attribute__((noinline)) MsEvalRet ms_eval_dielectric_call(Sampler smp, f3 wi, f3 wo, bool wo_outside, f2 alpha, Ior ext_ior, Ior int_ior, ThinfilmEval tf) {
  const f3 v = ms_eval_dielectric_inline(smp, wi, wo, wo_outside, alpha, ext_ior, int_ior, tf);
  return {v, smp.seed};
}
ETX_DEV f3 ms_eval_dielectric(Sampler& smp, const f3& wi, const f3& wo, bool wo_outside, const f2 alpha, const Ior& ext_ior, const Ior& int_ior, const ThinfilmEval& tf) {
  const MsEvalRet r = ms_eval_dielectric_call(smp, wi, wo, wo_outside, alpha, ext_ior, int_ior, tf);
  smp.seed = r.seed;
  return r.value;
}

// ---------------------------------------------------------------------------------------------------------------
// DielectricBSDF, bsdf_dielectric.hxx:61-259

ETX_DEV bool dielectric_is_delta(const DScene& s, const etx_abi_material& m, const f2 tex) {  // :251-254
  f2 r = evaluate_roughness(s, m, tex);
  return fmaxf(r.x, r.y) <= kDeltaAlphaTreshold;
}

ETX_DEV float dielectric_pdf(const DScene& s, const BsdfData& d, const f3& in_w_o, const etx_abi_material& m, Sampler& smp) {  // :198-249
  const Frame frame = {d.tan, d.btn, d.nrm, false};
  const f3 w_i = frame.to_local(-d.w_i);
  if (fabsf(w_i.z) <= kEpsilon)
    return 0.0f;
  const f3 w_o = frame.to_local(in_w_o);
  if (fabsf(w_o.z) <= kEpsilon)
    return 0.0f;
  const f2 roughness = evaluate_roughness(s, m, d.tex);
  const Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const bool outside = w_i.z > 0.0f;
  const bool reflection = w_i.z * w_o.z > 0.0f;
  f3 wh;
  float dwh_dwo;
  if (reflection) {
    wh = normalize(w_o + w_i);
    dwh_dwo = 1.0f / (4.0f * dot(w_o, wh));
  } else {
    const float eta = outside ? ior_eta_ratio(int_ior, ext_ior) : ior_eta_ratio(ext_ior, int_ior);
    wh = normalize(w_i + w_o * eta);
    const float sqrt_denom = dot(w_i, wh) + eta * dot(w_o, wh);
    dwh_dwo = sqr(eta) * dot(w_o, wh) / sqr(sqrt_denom);
  }
  wh *= (wh.z >= 0.0f) ? 1.0f : -1.0f;
  const MsRay ray = ms_ray(w_i * (outside ? 1.0f : -1.0f), roughness);
  const float d_ggx = D_ggx(wh, roughness);
  float prob = fmaxf(0.0f, dot(wh, ray.w) * d_ggx / ((1.0f + ray.Lambda) * ray.w.z));
  const float f = luminance(fresnel_calculate_g(dot(w_i, wh), outside ? ext_ior : int_ior, outside ? int_ior : ext_ior, tf));
  prob *= reflection ? f : (1.0f - f);
  return fabsf(prob * dwh_dwo) + fabsf(w_o.z);
}

ETX_DEV BsdfSample dielectric_sample(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {  // :74-147
  const Frame frame = {d.tan, d.btn, d.nrm, false};
  const f3 w_i = frame.to_local(-d.w_i);
  const bool in_outside = w_i.z > 0.0f;
  const float direction_scale = in_outside ? 1.0f : -1.0f;
  const Ior ext_ior = in_outside ? evaluate_refractive_index(s, m.ext_ior, d.wavelength) : evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const Ior int_ior = in_outside ? evaluate_refractive_index(s, m.int_ior, d.wavelength) : evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  BsdfSample r = sample_zero();
  r.weight = mk3(1.0f);
  const f2 roughness = evaluate_roughness(s, m, d.tex);
  MsRay ray = ms_ray(-direction_scale * w_i, roughness);
  ray.update_height(1.0f);
  bool ray_outside = true;
  uint32_t order = 0;
  while (true) {
    const float sampled_height = ms_sample_height(ray, smp.next());
    if (sampled_height == kMaxFloat)
      break;
    ray.update_height(sampled_height);
    const f2 rnd_slope = ((order == 0) && smp.has_fixed()) ? f2{smp.fixed_u, smp.fixed_v} : smp.next_2d();
    const float rnd_reflection = ((order == 0) && smp.has_fixed()) ? smp.fixed_w : smp.next();
    const MsDielectricSample next = ms_sample_phase_dielectric(rnd_slope, rnd_reflection, -ray.w, roughness, ray_outside ? ext_ior : int_ior, ray_outside ? int_ior : ext_ior, tf);
    r.weight *= next.weight;
    if (next.reflection) {
      ray.update_direction(next.w_o, roughness);
      ray.update_height(ray.h);
    } else {
      ray_outside = !ray_outside;
      ray.update_direction(-next.w_o, roughness);
      ray.update_height(-ray.h);
    }
    if (order++ > kScatteringOrderMax)
      return sample_zero();
  }
  f3 local_w_o = (ray_outside ? ray.w : -ray.w) * direction_scale;
  const uint32_t delta_sample = dielectric_is_delta(s, m, d.tex) ? kSampleDelta : 0u;
  if (w_i.z * local_w_o.z > 0.0f) {
    r.eta = 1.0f;
    r.weight = (r.weight / luminance(r.weight)) * apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
    r.properties = kSampleReflection | delta_sample;
    r.medium_index = d.medium;
  } else {
    const float eta = ior_eta_ratio(int_ior, ext_ior);
    r.eta = eta;
    r.weight = (r.weight / luminance(r.weight)) * apply_image(s, m.scattering, d.tex, nullptr, d.wavelength) * sqr(1.0f / eta);
    r.properties = kSampleTransmission | kSampleMediumChanged | delta_sample;
    r.medium_index = in_outside ? m.int_medium : m.ext_medium;
  }
  r.w_o = normalize(frame.from_local(local_w_o));
  r.pdf = dielectric_pdf(s, d, r.w_o, m, smp);
  return r;
}

ETX_DEV BsdfEval dielectric_evaluate(const DScene& s, const BsdfData& d, const f3& in_w_o, const etx_abi_material& m, Sampler& smp) {  // :149-196
  const Frame frame = {d.tan, d.btn, d.nrm, false};
  const f3 w_i = frame.to_local(-d.w_i);
  if (fabsf(w_i.z) <= kEpsilon)
    return eval_zero();
  const f3 w_o = frame.to_local(in_w_o);
  if (fabsf(w_o.z) <= kEpsilon)
    return eval_zero();
  const f2 roughness = evaluate_roughness(s, m, d.tex);
  const Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const bool forward_path = d.path_source == kPathCamera;
  const float backward_scale = fabsf(1.0f / w_i.z);
  // bsdf_dielectric.hxx:166-182: eight argument patterns of eval_dielectric, selected here so that the walk is ONE call
  // site: (a, b) = the two directions in the order the pattern passes them, the media swap for patterns that look from inside
  const bool i_up = w_i.z > 0.0f;
  const bool same_side = i_up ? (w_o.z >= 0.0f) : (w_o.z <= 0.0f);
  f3 a = w_i, b = w_o;
  bool swap_media = false, scaled = false;
  if (forward_path) {  // (w_i, w_o) seen from the side of w_i
    a = i_up ? w_i : -w_i;
    b = i_up ? w_o : -w_o;
    swap_media = (i_up == false);
  } else {             // adjoint: the roles of the directions swap, the walk starts from the side of w_o
    scaled = true;
    const bool o_up = same_side ? i_up : (i_up == false);  // side of w_o
    // patterns: (w_o, w_i) when w_o is above the surface, (-w_o, -w_i) otherwise
    a = o_up ? w_o : -w_o;
    b = o_up ? w_i : -w_i;
    swap_media = (o_up == false);
  }
  const Ior& first_ior = swap_media ? int_ior : ext_ior;
  const Ior& second_ior = swap_media ? ext_ior : int_ior;
  f3 value = ms_eval_dielectric(smp, a, b, same_side, roughness, first_ior, second_ior, tf);
  if (scaled)
    value = value * backward_scale;
  if (is_zero_rgb(value))
    return eval_zero();
  const bool reflection = w_i.z * w_o.z > 0.0f;
  BsdfEval e;
  e.eta = 1.0f;
  e.func = (2.0f * value) * apply_image(s, reflection ? m.reflectance : m.scattering, d.tex, nullptr, d.wavelength);
  e.bsdf = e.func * fabsf(w_o.z);
  e.pdf = dielectric_pdf(s, d, in_w_o, m, smp);
  return e;
}

// ---------------------------------------------------------------------------------------------------------------
// ThinfilmBSDF, bsdf_dielectric.hxx:3-59 (always delta: evaluate / pdf are zero)
ETX_DEV BsdfSample thinfilm_sample(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {
  const Frame frame = normal_frame(d);
  const Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const f3 fr = fresnel_calculate_g(dot(d.w_i, d.nrm), ext_ior, int_ior, tf);
  const float f = luminance(fr);
  BsdfSample r = sample_zero();
  if (smp.next() <= f) {
    r.w_o = normalize(reflect(d.w_i, frame.nrm));
    r.pdf = f;
    r.weight = apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength) * (fr / f);
    r.properties = kSampleDelta | kSampleReflection;
    r.medium_index = d.medium;
  } else {
    r.w_o = d.w_i;
    r.pdf = 1.0f - f;
    r.weight = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength) * ((mk3(1.0f) - fr) / (1.0f - f));
    r.properties = kSampleDelta | kSampleTransmission | kSampleMediumChanged;
    r.medium_index = frame.entering ? m.int_medium : m.ext_medium;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// rough diffuse: microfacet random walk (diffuse_variation 1) and vMF diffuse (2), bsdf_external.hxx:177-205, 557-894

ETX_DEV f3 ms_sample_vndf(Sampler& smp, const f3& wi, const f2 alpha) {  // :177-205
  const f3 wi_11 = normalize(f3{alpha.x * wi.x, alpha.y * wi.y, wi.z});
  f2 slope_11 = ms_sample_p22_11(acosf(wi_11.z), smp.next_2d());
  const float phi = atan2f(wi_11.y, wi_11.x);
  float sp, cp;
  sincosf(phi, &sp, &cp);
  f2 slope = {cp * slope_11.x - sp * slope_11.y, sp * slope_11.x + cp * slope_11.y};
  slope.x *= alpha.x;
  slope.y *= alpha.y;
  if ((slope.x != slope.x) || isinf(slope.x))
    return (wi.z > 0.0f) ? f3{0.0f, 0.0f, 1.0f} : normalize(f3{wi.x, wi.y, 0.0f});
  return normalize(f3{-slope.x, -slope.y, 1.0f});
}

ETX_DEV f3 ms_sample_phase_diffuse(Sampler& smp, const f3& wm) {  // :557-578
  const float r1 = 2.0f * smp.next() - 1.0f;
  const float r2 = 2.0f * smp.next() - 1.0f;
  float phi = 0.0f;
  const float r = (r1 * r1 > r2 * r2) ? r1 : r2;
  if (r1 * r1 > r2 * r2)
    phi = (kPi / 4.0f) * (r2 / r1);
  else if ((r1 != 0.0f) && (r2 != 0.0f))
    phi = (kPi / 2.0f) - (r1 / r2) * (kPi / 4.0f);
  const float x = r * cosf(phi), y = r * sinf(phi);
  const float z = sqrtf(fmaxf(0.0f, 1.0f - x * x - y * y));
  const Basis b = orthonormal_basis(wm);
  return x * b.u + y * b.v + z * wm;
}

ETX_DEV f3 ms_eval_diffuse(Sampler& smp, const f3& wi, const f3& wo, const f2 alpha, const f3& albedo) {  // :580-629
  MsRay ray_shadowing = ms_ray(wo, alpha);
  MsRay ray = ms_ray(-wi, alpha);
  ray.update_height(1.0f);
  f3 res = mk3(0.0f), energy = mk3(1.0f);
  int order = 0;
  while (true) {
    ray.update_height(ms_sample_height(ray, smp.next()));
    if (ray.h == kMaxFloat)
      break;
    const f3 wm = ms_sample_vndf(smp, -ray.w, alpha);
    const f3 phase = energy * albedo * fmaxf(0.0f, dot(wm, wo) * kInvPi);
    if (order == 0) {
      const float G2_G1 = -ray.Lambda / (ray_shadowing.Lambda - ray.Lambda);
      if (G2_G1 > 0.0f)
        res += phase * G2_G1;
    } else {
      ray_shadowing.update_height(ray.h);
      res += phase * ray_shadowing.G1;
    }
    ray.update_direction(ms_sample_phase_diffuse(smp, wm), alpha);
    ray.update_height(ray.h);
    energy = energy * albedo;
    if ((order++ > int(kScatteringOrderMax)) || (ray.h != ray.h) || (ray.w.x != ray.w.x))
      return mk3(0.0f);
  }
  return res;
}

ETX_DEV f3 ms_sample_diffuse(Sampler& smp, const f3& wi, const f2 alpha, const f3& albedo, f3& energy) {  // :660-693
  energy = mk3(1.0f);
  MsRay ray = ms_ray(-wi, alpha);
  ray.update_height(1.0f);
  int order = 0;
  while (true) {
    ray.update_height(ms_sample_height(ray, smp.next()));
    if (ray.h == kMaxFloat)
      break;
    order++;
    const f3 wm = ms_sample_vndf(smp, -ray.w, alpha);
    ray.update_direction(ms_sample_phase_diffuse(smp, wm), alpha);
    ray.update_height(ray.h);
    energy = energy * albedo;
    if (order > int(kScatteringOrderMax)) {
      energy = mk3(0.0f);
      return f3{0.0f, 0.0f, 1.0f};
    }
  }
  return ray.w;
}

// VMF diffuse (d'Eon & Weidlich), bsdf_external.hxx:696-894
ETX_DEV float vmf_erf(float x) {  // :702-706
  const float e = expf(-x * x);
  const float kSqrtPI = 1.7724538509055159f;
  return (x >= 0.0f ? 1.0f : -1.0f) * 2.0f / kSqrtPI * sqrtf(1.0f - e) * (kSqrtPI / 2.0f + 31.0f / 200.0f * e - 341.0f / 8000.0f * e * e);
}
ETX_DEV float vmf_fm_channel(float ui, float uo, float r, float c) {  // fm, :718-723 (per colour channel)
  const float C = sqrtf(1.0f - c);
  const float Ck = (1.0f - 0.5441615108674713f * C - 0.45302863761693374f * (1.0f - c)) / (1.0f + 1.4293127703064865f * C);
  const float Ca = c / powf(1.0075f + 1.16942f * C, atanf((0.0225272f + (-0.264641f + r) * r) * vmf_erf(c)));
  return fmaxf(0.0f, 0.384016f * (-0.341969f + Ca) * Ca * Ck * (-0.0578978f / (0.287663f + ui * uo) + fabsf(-0.0898863f + tanhf(r))));
}
ETX_DEV float vmf_sigma_beckmann_expanded(float u, float m) {  // :725-738
  if (0.0f == m)
    return (u + fabsf(u)) / 2.0f;
  const float m2 = m * m;
  if (1.0f == u)
    return 1.0f - 0.5f * m2;
  const float expansion = -0.25f * m2 * (u + fabsf(u));
  const float u2 = u * u;
  return ((expf(u2 / (m2 * (-1.0f + u2))) * m * sqrtf(1.0f - u2)) / sqrtf(kPi) + u * (1.0f + vmf_erf(u / (m * sqrtf(1.0f - u2))))) / 2.0f + expansion;
}
ETX_DEV float vmf_coth(float x) {  // :740-742
  return (expf(-x) + expf(x)) / (-expf(-x) + expf(x));
}
ETX_DEV float vmf_sigma(float u, float m) {  // sigmaVMF, :745-786
  if (m < 0.25f)
    return vmf_sigma_beckmann_expanded(u, m);
  const float m2 = m * m, m4 = m2 * m2, m8 = m4 * m4;
  const float u2 = u * u, u4 = u2 * u2, u6 = u2 * u4, u8 = u4 * u4, u10 = u6 * u4, u12 = u6 * u6;
  const float coth2m2 = vmf_coth(2.0f / m2);
  const float sinh2m2 = sinhf(2.0f / m2);
  if (m > 0.9f)
    return 0.25f - 0.25f * u * (m2 - 2.0f * coth2m2) + 0.0390625f * (-1.0f + 3.0f * u2) * (4.0f + 3.0f * m4 - 6.0f * m2 * coth2m2);
  const float q2 = 1.0132789611816406e-6f * (35.0f - 1260.0f * u2 + 6930.0f * u4 - 12012.0f * u6 + 6435.0f * u8) * (1.0f + coth2m2) *
                   (-256.0f - 315.0f * m4 * (128.0f + 33.0f * m4 * (80.0f + 364.0f * m4 + 195.0f * m8)) + 18.0f * m2 * (256.0f + 385.0f * m4 * (32.0f + 312.0f * m4 + 585.0f * m8)) * coth2m2) * sinh2m2;
  const float q1 = 9.12696123123169e-8f * (-63.0f + 3465.0f * u2 - 30030.0f * u4 + 90090.0f * u6 - 109395.0f * u8 + 46189.0f * u10) * (1.0f + coth2m2) *
                   (-1024.0f - 495.0f * m4 * (768.0f + 91.0f * m4 * (448.0f + 15.0f * m4 * (448.0f + 1836.0f * m4 + 969.0f * m8))) +
                     110.0f * m2 * (256.0f + 117.0f * m4 * (256.0f + 21.0f * m4 * (336.0f + 85.0f * m4 * (32.0f + 57.0f * m4)))) * coth2m2) * sinh2m2;
  const float q0 = 4.3655745685100555e-9f * (231.0f - 18018.0f * u2 + 225225.0f * u4 - 1.02102e6f * u6 + 2.078505e6f * u8 - 1.939938e6f * u10 + 676039.0f * u12) * (1.0f + coth2m2) *
                   (-4096.0f - 3003.0f * m4 * (1024.0f + 45.0f * m4 * (2560.0f + 51.0f * m4 * (1792.0f + 285.0f * m4 * (80.0f + 308.0f * m4 + 161.0f * m8)))) +
                     78.0f * m2 * (2048.0f + 385.0f * m4 * (1280.0f + 153.0f * m4 * (512.0f + 57.0f * m4 * (192.0f + 35.0f * m4 * (40.0f + 69.0f * m4))))) * coth2m2) * sinh2m2;
  return 0.25f - 0.25f * u * (m2 - 2.0f * coth2m2) + 0.0390625f * (-1.0f + 3.0f * u2) * (4.0f + 3.0f * m4 - 6.0f * m2 * coth2m2) -
         0.000732421875f * (3.0f - 30.0f * u2 + 35.0f * u4) * (16.0f + 180.0f * m4 + 105.0f * m8 - 10.0f * m2 * (8.0f + 21.0f * m4) * coth2m2) +
         0.000049591064453125f * (-5.0f + 105.0f * u2 - 315.0f * u4 + 231.0f * u6) * (64.0f + 105.0f * m4 * (32.0f + 180.0f * m4 + 99.0f * m8) - 42.0f * m2 * (16.0f + 240.0f * m4 + 495.0f * m8) * coth2m2) +
         (q2 / expf(2.0f / m2)) - (q1 / expf(2.0f / m2)) + (q0 / expf(2.0f / m2));
}

ETX_DEV f3 vmf_diffuse_brdf(const f3& w_i, const f3& w_o, const f2 roughness, const f3& albedo) {  // vMFdiffuseBRDF, :788-894
  const float r = fminf(fmaxf(sqrtf(roughness.x * roughness.y), 0.0f), 1.0f - 4.0f * kEpsilon);
  if (r == 0.0f)
    return albedo * kInvPi;
  const float cos_theta_i = w_i.z, sin_theta_i = sqrtf(1.0f - cos_theta_i * cos_theta_i);
  const float cos_theta_o = w_o.z, sin_theta_o = sqrtf(1.0f - cos_theta_o * cos_theta_o);
  float cos_phi_diff = 0.0f;
  if (sin_theta_i > 0.0f && sin_theta_o > 0.0f) {
    const float sin_phi_i = fminf(fmaxf(w_i.y / sin_theta_i, -1.0f), 1.0f), cos_phi_i = fminf(fmaxf(w_i.x / sin_theta_i, -1.0f), 1.0f);
    const float sin_phi_o = fminf(fmaxf(w_o.y / sin_theta_o, -1.0f), 1.0f), cos_phi_o = fminf(fmaxf(w_o.x / sin_theta_o, -1.0f), 1.0f);
    cos_phi_diff = fminf(fmaxf(cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o, -1.0f), 1.0f);
  }
  const float phi = acosf(cos_phi_diff);
  const float ui = w_i.z, uo = w_o.z;
  const float m = -logf(1.0f - sqrtf(r));
  const float sigmai = vmf_sigma(ui, m), sigmao = vmf_sigma(uo, m), sigmano = vmf_sigma(-uo, m);
  const float sigio = sigmai * sigmao;
  const float sigdenom = uo * sigmai + ui * sigmano;
  const float r2 = r * r, r25 = r2 * sqrtf(r), r3 = r * r2, r4 = r2 * r2, r45 = r4 * sqrtf(r), r5 = r3 * r2;
  const float ui2 = saturate(ui * ui), uo2 = saturate(uo * uo);
  const float sqrtuiuo = sqrtf((1.0f - ui2) * (1.0f - uo2));
  const float C100 = 1.0f + (-0.1f * r + 0.84f * r4) / (1.0f + 9.0f * r3);
  const float C101 = (0.0173f * r + 20.4f * r2 - 9.47f * r3) / (1.0f + 7.46f * r);
  const float C102 = (-0.927f * r + 2.37f * r2) / (1.24f + r2);
  const float C103 = (-0.110f * r - 1.54f * r2) / (1.0f - 1.05f * r + 7.1f * r2);
  const float f10 = ((C100 + C101 * ui * uo + C102 * ui2 * uo2 + C103 * (ui2 + uo2)) * sigio) / sigdenom;
  const float C110 = (0.54f * r - 0.182f * r3) / (1.0f + 1.32f * r2);
  const float C111 = (-0.097f * r + 0.62f * r2 - 0.375f * r3) / (1.0f + 0.4f * r3);
  const float C112 = 0.283f + 0.862f * r - 0.681f * r2;
  const float f11 = (sqrtuiuo * (C110 + C111 * ui * uo)) * powf(sigio, C112) / sigdenom;
  const float C120 = (2.25f * r + 5.1f * r2) / (1.0f + 9.8f * r + 32.4f * r2);
  const float C121 = (-4.32f * r + 6.0f * r3) / (1.0f + 9.7f * r + 287.0f * r3);
  const float f12 = ((1.0f - ui2) * (1.0f - uo2) * (C120 + C121 * uo) * (C120 + C121 * ui)) / (ui + uo);
  const float C200 = (0.00056f * r + 0.226f * r2) / (1.0f + 7.07f * r2);
  const float C201 = (-0.268f * r + 4.57f * r2 - 12.04f * r3) / (1.0f + 36.7f * r3);
  const float C202 = (0.418f * r + 2.52f * r2 - 0.97f * r3) / (1.0f + 10.0f * r2);
  const float C203 = (0.068f * r - 2.25f * r2 + 2.65f * r3) / (1.0f + 21.4f * r3);
  const float C204 = (0.050f * r - 4.22f * r3) / (1.0f + 17.6f * r2 + 43.1f * r3);
  const float f20 = (C200 + C201 * ui * uo + C203 * ui2 * uo2 + C202 * (ui + uo) + C204 * (ui2 + uo2)) / (ui + uo);
  const float C210 = (-0.049f * r - 0.027f * r3) / (1.0f + 3.36f * r2);
  const float C211 = (2.77f * r2 - 8.332f * r25 + 6.073f * r3) / (1.0f + 50.0f * r4);
  const float C212 = (-0.431f * r2 - 0.295f * r3) / (1.0f + 23.9f * r3);
  const float f21 = (sqrtuiuo * (C210 + C211 * ui * uo + C212 * (ui + uo))) / (ui + uo);
  const float C300 = (-0.083f * r3 + 0.262f * r4) / (1.0f - 1.9f * r2 + 38.6f * r4);
  const float C301 = (-0.627f * r2 + 4.95f * r25 - 2.44f * r3) / (1.0f + 31.5f * r4);
  const float C302 = (0.33f * r2 + 0.31f * r25 + 1.4f * r3) / (1.0f + 20.0f * r3);
  const float C303 = (-0.74f * r2 + 1.77f * r25 - 4.06f * r3) / (1.0f + 215.0f * r5);
  const float C304 = (-1.026f * r3) / (1.0f + 5.81f * r2 + 13.2f * r3);
  const float f30 = (C300 + C301 * ui * uo + C303 * ui2 * uo2 + C302 * (ui + uo) + C304 * (ui2 + uo2)) / (ui + uo);
  const float C310 = (0.028f * r2 - 0.0132f * r3) / (1.0f + 7.46f * r2 - 3.315f * r4);
  const float C311 = (-0.134f * r2 + 0.162f * r25 + 0.302f * r3) / (1.0f + 57.5f * r45);
  const float C312 = (-0.119f * r2 + 0.5f * r25 - 0.207f * r3) / (1.0f + 18.7f * r3);
  const float f31 = (sqrtuiuo * (C310 + C311 * ui * uo + C312 * (ui + uo))) / (ui + uo);
  const f3 t0 = albedo * fmaxf(0.0f, f10 + f11 * cosf(phi) * 2.0f + f12 * cosf(2.0f * phi) * 2.0f);
  const f3 t1 = albedo * albedo * fmaxf(0.0f, f20 + f21 * cosf(phi) * 2.0f);
  const f3 t2 = albedo * albedo * albedo * fmaxf(0.0f, f30 + f31 * cosf(phi) * 2.0f);
  const f3 t4 = {vmf_fm_channel(ui, uo, r, albedo.x), vmf_fm_channel(ui, uo, r, albedo.y), vmf_fm_channel(ui, uo, r, albedo.z)};
  return (t0 + t1 + t2) * kInvPi + t4;
}

// ---------------------------------------------------------------------------------------------------------------
// PlasticBSDF, bsdf_plastic.hxx:3-186

ETX_DEV f3 ggx_sample_normal(const Frame& frame, const f2 in_alpha, Sampler& smp, const f3& in_w_i) {  // NormalDistribution::sample, bsdf.hxx:128-146
  const float kMinAlpha = 1.0f / 256.0f;
  const f2 alpha = {fmaxf(kMinAlpha, in_alpha.x), fmaxf(kMinAlpha, in_alpha.y)};
  const f3 w_i = frame.to_local(-in_w_i);
  const f3 v_h = normalize(f3{alpha.x * w_i.x, alpha.y * w_i.y, w_i.z});
  const float v_h_len = v_h.x * v_h.x + v_h.y * v_h.y;
  const f3 u = v_h_len > 0.0f ? f3{-v_h.y, v_h.x, 0.0f} / sqrtf(v_h_len) : f3{1.0f, 0.0f, 0.0f};
  const f3 v = cross(v_h, u);
  const float r = sqrtf(smp.next());
  const float phi = kDoublePi * smp.next();
  const float t1 = r * cosf(phi);
  float t2 = r * sinf(phi);
  const float sc = 0.5f * (1.0f + v_h.z);
  t2 = (1.0f - sc) * sqrtf(1.0f - t1 * t1) + sc * t2;
  const f3 n_h = t1 * u + t2 * v + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * v_h;
  const f3 local_m = normalize(f3{alpha.x * n_h.x, alpha.y * n_h.y, n_h.z});
  return frame.from_local(local_m);
}

ETX_DEV f3 plastic_specular_func(const DScene& s, const BsdfData& d, const f3& in_w_o, const etx_abi_material& m, Sampler& smp) {  // :14-35
  const Frame frame = {d.tan, d.btn, d.nrm, false};
  const f3 w_i = frame.to_local(-d.w_i);
  if (w_i.z <= kEpsilon)
    return mk3(0.0f);
  const f3 w_o = frame.to_local(in_w_o);
  if (w_o.z <= kEpsilon)
    return mk3(0.0f);
  const f2 roughness = evaluate_roughness(s, m, d.tex);
  const Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const f3 value = ms_eval_dielectric(smp, w_i, w_o, true, roughness, ext_ior, int_ior, tf);
  return 2.0f * value * apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
}

ETX_DEV float plastic_specular_pdf(const DScene& s, const BsdfData& d, const f3& in_w_o, const etx_abi_material& m, Sampler& smp) {  // :37-74
  const Frame frame = {d.tan, d.btn, d.nrm, false};
  const f3 w_i = frame.to_local(-d.w_i);
  if (w_i.z <= kEpsilon)
    return 0.0f;
  const f3 w_o = frame.to_local(in_w_o);
  if (w_o.z <= kEpsilon)
    return 0.0f;
  const Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const f2 roughness = evaluate_roughness(s, m, d.tex);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const f3 wh = normalize(w_o + w_i);
  const float dwh_dwo = 1.0f / (4.0f * dot(w_o, wh));
  const MsRay ray = ms_ray(w_i, roughness);
  const float d_ggx = D_ggx(wh, roughness);
  float prob = fmaxf(0.0f, dot(wh, ray.w) * d_ggx / ((1.0f + ray.Lambda) * ray.w.z));
  prob *= luminance(fresnel_calculate_g(dot(w_i, wh), ext_ior, int_ior, tf));
  return fabsf(prob * dwh_dwo);
}

ETX_DEV BsdfEval diffuse_layer_v(const DScene& s, const BsdfData& d, const f3& local_w_i, const f3& local_w_o, const etx_abi_material& m, Sampler& smp);

ETX_DEV BsdfEval plastic_evaluate(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m, Sampler& smp) {  // :117-152
  const Frame frame = normal_frame(d);
  const f3 mh = normalize(w_o - d.w_i);
  const float n_dot_o = dot(frame.nrm, w_o);
  const float m_dot_o = dot(mh, w_o);
  if ((n_dot_o <= kEpsilon) || (m_dot_o <= kEpsilon))
    return eval_zero();
  const Ior eta_e = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior eta_i = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const f3 fr = fresnel_calculate_g(dot(d.w_i, mh), eta_e, eta_i, tf);
  const f3 local_w_i = frame.to_local(-d.w_i);
  const f3 local_w_o = frame.to_local(w_o);
  const BsdfEval diff_layer = diffuse_layer_v(s, d, local_w_i, local_w_o, m, smp);
  const f3 spec_layer = plastic_specular_func(s, d, w_o, m, smp);
  const float spec_pdf = plastic_specular_pdf(s, d, w_o, m, smp);
  BsdfEval e;
  e.eta = 1.0f;
  e.func = diff_layer.func * (mk3(1.0f) - fr) + spec_layer / n_dot_o;
  e.bsdf = diff_layer.func * (mk3(1.0f) - fr) * n_dot_o + spec_layer;
  e.pdf = diff_layer.pdf * luminance(mk3(1.0f) - fr) + spec_pdf;
  return e;
}

ETX_DEV float plastic_pdf(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m, Sampler& smp) {  // :154-176
  const Frame frame = normal_frame(d);
  const f3 mh = normalize(w_o - d.w_i);
  const float m_dot_o = dot(mh, w_o);
  const float n_dot_o = dot(frame.nrm, w_o);
  if ((n_dot_o <= kEpsilon) || (m_dot_o <= kEpsilon))
    return 0.0f;
  const Ior eta_e = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior eta_i = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const f3 fr = fresnel_calculate_g(dot(d.w_i, mh), eta_e, eta_i, tf);
  const float diff_pdf = kInvPi * n_dot_o;
  const float spec_pdf = plastic_specular_pdf(s, d, w_o, m, smp);
  return diff_pdf * luminance(mk3(1.0f) - fr) + spec_pdf;
}

ETX_DEV BsdfSample plastic_sample(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {  // :76-115
  const Frame frame = normal_frame(d);
  const f2 roughness = evaluate_roughness(s, m, d.tex);
  const f3 mh = ggx_sample_normal(frame, roughness, smp, d.w_i);
  const Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  const Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  const f3 f = fresnel_calculate_g(dot(d.w_i, mh), ext_ior, int_ior, tf);
  const f3 w_i = frame.to_local(-d.w_i);
  if (w_i.z <= kEpsilon)
    return sample_zero();
  f3 in_w_o = mk3(0.0f);
  bool sample_diffuse = smp.next() > luminance(f);
  if (sample_diffuse == false) {
    in_w_o = reflect(d.w_i, mh);
    sample_diffuse = dot(frame.nrm, in_w_o) <= kEpsilon;
  }
  if (sample_diffuse)
    in_w_o = frame.from_local(sample_cosine_distribution(smp.next_2d(), 1.0f));
  const BsdfEval eval = plastic_evaluate(s, d, in_w_o, m, smp);
  BsdfSample r = sample_zero();
  r.w_o = in_w_o;
  r.weight = eval.bsdf / eval.pdf;
  r.properties = kSampleReflection | (sample_diffuse ? kSampleDiffuse : 0u);
  r.medium_index = d.medium;
  r.pdf = eval.pdf;
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// VelvetBSDF, bsdf_velvet.hxx:3-126

ETX_DEV float velvet_lambda_l(float r, float x) {  // :29-42
  x = fmaxf(x, 0.0f);
  const float one_minus_r_sq = sqr(1.0f - r);
  auto lerp_x = [&](float a, float b) { return one_minus_r_sq * a + (1.0f - one_minus_r_sq) * b; };
  const float a = lerp_x(25.3245f, 21.5473f), b = lerp_x(3.32435f, 3.82987f), c = lerp_x(0.16801f, 0.19823f);
  const float dd = lerp_x(-1.27393f, -1.97760f), e = lerp_x(-4.85967f, -4.32054f);
  return a / (1.0f + b * powf(x, c)) + dd * x + e;
}
ETX_DEV float velvet_lambda(float r, float cos_t) {  // :44-49
  if (cos_t < 0.5f)
    return expf(velvet_lambda_l(r, cos_t));
  return expf(2.0f * velvet_lambda_l(r, 0.5f) - velvet_lambda_l(r, 1.0f - cos_t));
}
ETX_DEV float velvet_fresnel_approximate(float f0, float f90, float cos_t) {  // :51-53
  return f0 + (f90 - f0) * powf(fmaxf(1.0f - cos_t, 0.0f), 5.0f);
}

ETX_DEV BsdfEval velvet_evaluate(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m) {  // :62-108
  const Frame frame = normal_frame(d);
  const float n_dot_o = fmaxf(0.0f, dot(w_o, frame.nrm));
  const float n_dot_i = fmaxf(0.0f, -dot(d.w_i, frame.nrm));
  if ((n_dot_o <= kEpsilon) || (n_dot_i <= kEpsilon))
    return eval_zero();
  const f3 mh = normalize(w_o - d.w_i);
  const float m_dot_o = fmaxf(0.0f, dot(w_o, mh));
  const float m_dot_i = fmaxf(0.0f, -dot(d.w_i, mh));
  if ((m_dot_o <= kEpsilon) || (m_dot_i <= kEpsilon))
    return eval_zero();
  const f2 roughness = evaluate_roughness(s, m, d.tex);
  float specular_scale_base = 0.0f;
  const float alpha = 0.5f * (roughness.x + roughness.y);
  if (alpha > kEpsilon) {
    const float inv_alpha = 1.0f / (kEpsilon + alpha);
    const float m_dot_n = dot(mh, frame.nrm);
    const float sin_t = 1.0f - m_dot_n * m_dot_n;
    const float dd = (2.0f + inv_alpha) * powf(sin_t, 0.5f * inv_alpha) / kDoublePi;
    const float l_i = velvet_lambda(alpha, n_dot_i), l_o = velvet_lambda(alpha, n_dot_o);
    const float g = 1.0f / (1.0f + l_i + l_o);
    specular_scale_base = 0.25f * dd * g / n_dot_i;
  }
  const f3 diffuse = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
  const f3 specular = apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
  // diffuse_burley, :55-60
  const float f90 = 0.5f + 2.0f * alpha * m_dot_o * m_dot_o;
  const float diffuse_scale = velvet_fresnel_approximate(1.0f, f90, n_dot_o) * velvet_fresnel_approximate(1.0f, f90, n_dot_i) * kInvPi;
  BsdfEval e;
  e.eta = 1.0f;
  e.func = diffuse * diffuse_scale + specular * (specular_scale_base / n_dot_o);
  e.bsdf = diffuse * (diffuse_scale * n_dot_o) + specular * specular_scale_base;
  e.pdf = 1.0f / kDoublePi;
  return e;
}

ETX_DEV BsdfSample velvet_sample(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {  // :11-27
  const Frame frame = normal_frame(d);
  const f3 w_o = sample_cosine_distribution(smp.next_2d(), frame.nrm, 0.0f);
  const BsdfEval eval = velvet_evaluate(s, d, w_o, m);
  BsdfSample r = sample_zero();
  r.w_o = w_o;
  r.properties = kSampleReflection | kSampleDiffuse;
  r.medium_index = d.medium;
  r.eta = 1.0f;
  r.pdf = eval.pdf;
  r.weight = eval.bsdf / eval.pdf;
  return r;
}

ETX_DEV float velvet_pdf(const BsdfData& d) {  // :110-116
  return normal_frame(d).entering ? (1.0f / kDoublePi) : 0.0f;
}

}  // namespace etxd
