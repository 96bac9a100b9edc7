// dev_bsdf.h - scattering models (RGB mode: a spectral response is an f3 "integrated" value).
// Dispatch mirrors bsdf::sample/evaluate/pdf/reverse_pdf (sources/etx/render/shared/scene_bsdf.hxx:56-107).
// Implemented classes: Diffuse (variation 0), Mirror, Boundary, Void, Translucent, Conductor (Heitz multiple
// scattering random walk incl. the delta limit). Other classes are rejected by etx_hip_upload_scene with
// ETX_HIP_ERROR_UNSUPPORTED until they are restated here - never approximated.
#pragma once

#include "dev_scene.h"

namespace etxd {

enum : uint32_t { kPathCamera = 1u, kPathLight = 2u };  // bsdf.hxx:16-20 PathSource

struct BsdfData {  // bsdf.hxx:22-48 BSDFData (spectrum query dropped: RGB mode)
  f3 nrm, tan, btn;
  f2 tex;
  f3 w_i;
  uint32_t medium;
  uint32_t path_source;
  float wavelength;  // spectrum_sample: the path's wavelength in spectral mode (ignored in RGB mode)
};

ETX_DEV BsdfData make_bsdf_data(const Vtx& v, const f3& w_i, uint32_t medium, uint32_t path_source, float wavelength) {
  return BsdfData{v.nrm, v.tan, v.btn, v.tex, w_i, medium, path_source, wavelength};
}

struct Frame {  // math.hxx:614-646 LocalFrame
  f3 tan, btn, nrm;
  bool entering;
  ETX_DEV f3 to_local(const f3& v) const {
    return {dot(tan, v), dot(btn, v), dot(nrm, v)};
  }
  ETX_DEV f3 from_local(const f3& v) const {
    return tan * v.x + btn * v.y + nrm * v.z;
  }
};

ETX_DEV Frame normal_frame(const BsdfData& d) {  // bsdf.hxx:37-40 get_normal_frame
  bool entering = dot(d.nrm, d.w_i) < 0.0f;
  return entering ? Frame{d.tan, d.btn, d.nrm, true} : Frame{-d.tan, -d.btn, -d.nrm, false};
}

struct BsdfEval {  // bsdf.hxx:50-68
  f3 func, bsdf;
  float pdf, eta;
  ETX_DEV bool valid() const {
    return pdf > 0.0f;
  }
};
ETX_DEV BsdfEval eval_zero() {
  return {mk3(0.0f), mk3(0.0f), 0.0f, 1.0f};
}

enum : uint32_t {  // bsdf.hxx:71-77 BSDFSample::Properties
  kSampleDiffuse = 1u << 0,
  kSampleReflection = 1u << 1,
  kSampleTransmission = 1u << 2,
  kSampleMediumChanged = 1u << 3,
  kSampleDelta = 1u << 4,
};

struct BsdfSample {  // bsdf.hxx:70-118
  f3 weight;
  f3 w_o;
  float pdf, eta;
  uint32_t properties, medium_index;
  ETX_DEV bool valid() const {
    return pdf > 0.0f;
  }
  ETX_DEV bool is_delta() const {
    return (properties & kSampleDelta) != 0u;
  }
};
ETX_DEV BsdfSample sample_zero() {
  return {mk3(0.0f), mk3(0.0f), 0.0f, 1.0f, 0u, kInvalid};
}

// bsdf.hxx:232-239
ETX_DEV float fix_shading_normal(const f3& n_g, const f3& n_s, const f3& w_i, const f3& w_o) {
  float w_i_g = dot(w_i, n_g), w_i_s = dot(w_i, n_s), w_o_g = dot(w_o, n_g), w_o_s = dot(w_o, n_s);
  float den = fmaxf(kInvMaxHalf, fabsf(w_o_s * w_i_g));
  return fabsf(w_o_g * w_i_s) / den;
}

// ---------------------------------------------------------------------------------------------------------------
// complex helpers for fresnel (bsdf.hxx:241-377 uses std::complex<float>)
struct cplx {
  float re, im;
};
ETX_DEV cplx cmul(cplx a, cplx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
ETX_DEV cplx cadd(cplx a, cplx b) {
  return {a.re + b.re, a.im + b.im};
}
ETX_DEV cplx csub(cplx a, cplx b) {
  return {a.re - b.re, a.im - b.im};
}
ETX_DEV cplx cdiv(cplx a, cplx b) {  // complex_div_conj: a * conj(b) / norm(b), bsdf.hxx:243-247
  float denom = b.re * b.re + b.im * b.im;
  cplx num = cmul(a, cplx{b.re, -b.im});
  return {num.re / denom, num.im / denom};
}
ETX_DEV cplx csqrt(cplx z) {
  float r = sqrtf(z.re * z.re + z.im * z.im);
  if (r == 0.0f)
    return {0.0f, 0.0f};
  float a = sqrtf(fmaxf(0.0f, 0.5f * (r + z.re)));
  float b = sqrtf(fmaxf(0.0f, 0.5f * (r - z.re)));
  return {a, (z.im < 0.0f) ? -b : b};
}
ETX_DEV float cnorm(cplx z) {
  return z.re * z.re + z.im * z.im;
}

// bsdf.hxx:285-291 fresnel_generic (+ reflectance :249-266)
ETX_DEV float fresnel_generic(float cos_theta_i, cplx ext_ior, cplx int_ior) {
  cplx ratio = cdiv(ext_ior, int_ior);
  cplx r2 = cmul(ratio, ratio);
  float s = 1.0f - cos_theta_i * cos_theta_i;
  cplx sin2 = {r2.re * s, r2.im * s};
  cplx cos_o = csqrt(cplx{1.0f - sin2.re, -sin2.im});
  cplx cos_i = {cos_theta_i, 0.0f};
  if ((cos_i.re == 0.0f) && (cos_o.re == 0.0f) && (cos_o.im == 0.0f))
    return 1.0f;
  if ((ext_ior.re == int_ior.re) && (ext_ior.im == int_ior.im))
    return 0.0f;
  cplx ni_ci = cmul(ext_ior, cos_i), nj_cj = cmul(int_ior, cos_o);
  cplx nj_ci = cmul(int_ior, cos_i), ni_cj = cmul(ext_ior, cos_o);
  cplx rs = cdiv(csub(ni_ci, nj_cj), cadd(ni_ci, nj_cj));
  cplx rp = cdiv(csub(nj_ci, ni_cj), cadd(nj_ci, ni_cj));
  return 0.5f * (cnorm(rs) + cnorm(rp));
}

struct Ior {  // RefractiveIndex::Sample (spectrum.hxx:553-590)
  f3 eta, k;
  uint32_t cls;
  uint32_t spectral;  // values are single-wavelength samples (replicated), not RGB / XYZ integrals
};

ETX_DEV Ior evaluate_refractive_index(const DScene& s, const etx_abi_refractive_index& ri, float wavelength) {  // scene.hxx:311-317
  Ior r;
  r.cls = ri.cls;
  r.spectral = s.spectral;
  r.eta = (ri.eta_index == kInvalid) ? mk3(1.0f) : spectrum_eval(s, ri.eta_index, wavelength);
  r.k = (ri.k_index == kInvalid) ? mk3(0.0f) : spectrum_eval(s, ri.k_index, wavelength);
  return r;
}

constexpr uint32_t kSpectrumClassConductor = 2u;  // SpectralDistribution::Class::Conductor, spectrum.hxx:450-456
constexpr uint32_t kSpectrumClassDielectric = 3u;

ETX_DEV bool is_zero_rgb(const f3& v) {  // SpectralResponse::is_zero, spectrum.hxx:317-319
  return (v.x <= kEpsilon) && (v.y <= kEpsilon) && (v.z <= kEpsilon);
}

// Thinfilm::Eval (material.hxx:23-27) from evaluate_thinfilm (scene_bsdf.hxx:110-126). In RGB mode the three film
// wavelengths are jittered around 610 / 537 / 450 nm with three draws of the path's sampler.
struct ThinfilmEval {
  Ior ior;
  f3 rgb_wavelengths;
  float thickness;
};

ETX_DEV ThinfilmEval evaluate_thinfilm(const DScene& s, const etx_abi_thinfilm& film, const f2 uv, Sampler& smp, float wavelength) {
  ThinfilmEval r;
  r.ior.cls = 0u, r.ior.spectral = s.spectral, r.ior.eta = mk3(0.0f), r.ior.k = mk3(0.0f);
  r.rgb_wavelengths = {610.0f, 537.0f, 450.0f};
  r.thickness = 0.0f;
  if (film.max_thickness * film.min_thickness <= 0.0f)
    return r;
  const float t = (film.thickness_image == kInvalid) ? 1.0f : image_evaluate(s.images[film.thickness_image], uv, nullptr).x;
  r.thickness = film.min_thickness + (film.max_thickness - film.min_thickness) * t;  // lerp(min, max, t)
  if (s.spectral) {  // scene_bsdf.hxx:118: the film sees the path's wavelength, no draws
    r.rgb_wavelengths = mk3(wavelength);
  } else {
    r.rgb_wavelengths.x = 610.0f + 45.0f * (2.0f * smp.next() - 1.0f);
    r.rgb_wavelengths.y = 537.0f + 47.0f * (2.0f * smp.next() - 1.0f);
    r.rgb_wavelengths.z = 450.0f + 23.5f * (2.0f * smp.next() - 1.0f);
  }
  r.ior = evaluate_refractive_index(s, film.ior, wavelength);
  return r;
}

ETX_DEV cplx cscale(cplx a, float k) {
  return {a.re * k, a.im * k};
}
ETX_DEV cplx cexp(cplx z) {  // complex_exp
  const float e = expf(z.re);
  float sn, cs;
  sincosf(z.im, &sn, &cs);
  return {e * cs, e * sn};
}
struct FresnelPair {
  cplx s, p;
};
// bsdf.hxx:249-266 reflectance, :268-285 transmittance
ETX_DEV FresnelPair fresnel_reflectance(cplx ni, cplx cos_i, cplx nj, cplx cos_j) {
  if ((cos_i.re == 0.0f) && (cos_j.re == 0.0f) && (cos_i.im == 0.0f) && (cos_j.im == 0.0f))
    return {cplx{1.0f, 0.0f}, cplx{1.0f, 0.0f}};
  if ((ni.re == nj.re) && (ni.im == nj.im))
    return {cplx{0.0f, 0.0f}, cplx{0.0f, 0.0f}};
  const cplx ni_ci = cmul(ni, cos_i), nj_cj = cmul(nj, cos_j), nj_ci = cmul(nj, cos_i), ni_cj = cmul(ni, cos_j);
  return {cdiv(csub(ni_ci, nj_cj), cadd(ni_ci, nj_cj)), cdiv(csub(nj_ci, ni_cj), cadd(nj_ci, ni_cj))};
}
ETX_DEV FresnelPair fresnel_transmittance(cplx ni, cplx cos_i, cplx nj, cplx cos_j) {
  if ((cos_i.re == 0.0f) && (cos_j.re == 0.0f) && (cos_i.im == 0.0f) && (cos_j.im == 0.0f))
    return {cplx{0.0f, 0.0f}, cplx{0.0f, 0.0f}};
  if ((ni.re == nj.re) && (ni.im == nj.im))
    return {cplx{1.0f, 0.0f}, cplx{1.0f, 0.0f}};
  const cplx two_ni_ci = cscale(cmul(ni, cos_i), 2.0f);
  return {cdiv(two_ni_ci, cadd(cmul(ni, cos_i), cmul(nj, cos_j))), cdiv(two_ni_ci, cadd(cmul(ni, cos_j), cmul(nj, cos_i)))};
}

// bsdf.hxx:299-337 fresnel_thinfilm
ETX_DEV float fresnel_thinfilm(float wavelength, float cos_theta_0, cplx ext_ior, cplx film_ior, cplx int_ior, float thickness) {
  if (cos_theta_0 == 0.0f)
    return 0.0f;
  const cplx one = {1.0f, 0.0f};
  cplx ratio_01 = cdiv(ext_ior, film_ior);
  const cplx sin_theta_1_squared = cscale(cmul(ratio_01, ratio_01), 1.0f - cos_theta_0 * cos_theta_0);
  if (sin_theta_1_squared.re >= 1.0f)
    return 1.0f;
  const cplx cos_theta_1 = csqrt(csub(one, sin_theta_1_squared));
  cplx ratio_12 = cdiv(film_ior, int_ior);
  const cplx sin_theta_2_squared = cmul(cmul(ratio_12, ratio_12), csub(one, cmul(cos_theta_1, cos_theta_1)));
  if (sin_theta_2_squared.re >= 1.0f)
    return 1.0f;
  const cplx cos_theta_2 = csqrt(csub(one, sin_theta_2_squared));
  const cplx cos_0 = {cos_theta_0, 0.0f};
  const cplx ratio = cdiv(cmul(int_ior, cos_theta_2), cmul(ext_ior, cos_0));
  const float delta_10 = ext_ior.re < film_ior.re ? kPi : 0.0f;
  const float delta_21 = film_ior.re < int_ior.re ? kPi : 0.0f;
  const float phase_shift = delta_10 + delta_21;
  const FresnelPair r01 = fresnel_reflectance(ext_ior, cos_0, film_ior, cos_theta_1);
  const FresnelPair t01 = fresnel_transmittance(ext_ior, cos_0, film_ior, cos_theta_1);
  const FresnelPair r12 = fresnel_reflectance(film_ior, cos_theta_1, int_ior, cos_theta_2);
  const FresnelPair t12 = fresnel_transmittance(film_ior, cos_theta_1, int_ior, cos_theta_2);
  const cplx phi = cscale(cadd(cscale(cos_theta_1, kDoublePi * 2.0f * thickness), cscale(film_ior, phase_shift)), 1.0f / wavelength);
  const cplx exp_i_phi = cexp(cplx{-phi.im, phi.re});  // exp(i * phi)
  cplx tp = cdiv(cmul(t01.p, t12.p), csub(one, cmul(cmul(r01.p, r12.p), exp_i_phi)));
  tp = cmul(tp, tp);
  cplx ts = cdiv(cmul(t01.s, t12.s), csub(one, cmul(cmul(r01.s, r12.s), exp_i_phi)));
  ts = cmul(ts, ts);
  const cplx v = csub(one, cscale(cmul(ratio, cadd(tp, ts)), 0.5f));
  return sqrtf(v.re * v.re + v.im * v.im);  // complex_abs
}

// bsdf.hxx:337-375 fresnel::calculate, RGB branch
ETX_DEV f3 fresnel_calculate(float cos_theta, const Ior& ext_ior, const Ior& int_ior, const ThinfilmEval& tf) {
  cos_theta = fabsf(cos_theta);
  f3 values;
  if ((tf.thickness == 0.0f) || is_zero_rgb(tf.ior.eta)) {
    values.x = fresnel_generic(cos_theta, cplx{ext_ior.eta.x, ext_ior.k.x}, cplx{int_ior.eta.x, int_ior.k.x});
    values.y = fresnel_generic(cos_theta, cplx{ext_ior.eta.y, ext_ior.k.y}, cplx{int_ior.eta.y, int_ior.k.y});
    values.z = fresnel_generic(cos_theta, cplx{ext_ior.eta.z, ext_ior.k.z}, cplx{int_ior.eta.z, int_ior.k.z});
    if ((int_ior.cls == kSpectrumClassConductor) && (int_ior.spectral == 0u)) {
      // conductor IORs are stored as XYZ (spectrum.cxx:390-391); spectrum.hxx:142-148 xyz_to_rgb, :449 kRGBLuminanceScale
      f3 rgb = {
        3.24045420f * values.x - 1.5371385f * values.y - 0.4985314f * values.z,
        -0.9692660f * values.x + 1.8760108f * values.y + 0.0415560f * values.z,
        0.05564340f * values.x - 0.2040259f * values.y + 1.0572252f * values.z,
      };
      values = rgb * f3{0.817660332f, 1.05418909f, 1.09945524f};
    }
  } else {
    values.x = fresnel_thinfilm(tf.rgb_wavelengths.x, cos_theta, cplx{ext_ior.eta.x, ext_ior.k.x}, cplx{tf.ior.eta.x, tf.ior.k.x}, cplx{int_ior.eta.x, int_ior.k.x}, tf.thickness);
    values.y = fresnel_thinfilm(tf.rgb_wavelengths.y, cos_theta, cplx{ext_ior.eta.y, ext_ior.k.y}, cplx{tf.ior.eta.y, tf.ior.k.y}, cplx{int_ior.eta.y, int_ior.k.y}, tf.thickness);
    values.z = fresnel_thinfilm(tf.rgb_wavelengths.z, cos_theta, cplx{ext_ior.eta.z, ext_ior.k.z}, cplx{tf.ior.eta.z, tf.ior.k.z}, cplx{int_ior.eta.z, int_ior.k.z}, tf.thickness);
  }
  return {saturate(values.x), saturate(values.y), saturate(values.z)};
}

// The general classes call the Fresnel term from many places (every step of the Heitz walks); one real function per
// translation unit keeps the thin-film complex arithmetic out of each of them (dev_bsdf_ool.h explains the by-value ABI).
// The simple-material kernels keep the inline version above: with thinfilm_none() the thin-film branch folds away.
static __device__ __attribute__((noinline)) f3 fresnel_calculate_g(float cos_theta, Ior ext_ior, Ior int_ior, ThinfilmEval tf) {
  return fresnel_calculate(cos_theta, ext_ior, int_ior, tf);
}

ETX_DEV ThinfilmEval thinfilm_none() {
  ThinfilmEval r;
  r.ior.cls = 0u, r.ior.spectral = 0u, r.ior.eta = mk3(0.0f), r.ior.k = mk3(0.0f);
  r.rgb_wavelengths = {610.0f, 537.0f, 450.0f};
  r.thickness = 0.0f;
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Heitz et al. multiple-scattering microfacet model, conductor part (bsdf_external.hxx:16-353)
constexpr uint32_t kScatteringOrderMax = 16u;

struct MsRay {  // bsdf_external.hxx:16-70 RayInfo
  f3 w;
  float Lambda, h, C1, G1;
  ETX_DEV void update_direction(const f3& in_w, const f2 alpha) {
    w = in_w;
    if (w.z > 0.9999f) {
      Lambda = 0.0f;
      return;
    }
    if (w.z < -0.9999f) {
      Lambda = -1.0f;
      return;
    }
    const float theta = acosf(w.z);
    const float tan_theta = sinf(theta) / w.z;
    const float inv_sin2 = 1.0f / (1.0f - w.z * w.z);
    const float cos_phi2 = w.x * w.x * inv_sin2;
    const float sin_phi2 = w.y * w.y * inv_sin2;
    const float alpha_value = sqrtf(cos_phi2 * alpha.x * alpha.x + sin_phi2 * alpha.y * alpha.y);
    const float a = 1.0f / tan_theta / alpha_value;
    Lambda = 0.5f * (-1.0f + ((a > 0) ? 1.0f : -1.0f) * sqrtf(1.0f + 1.0f / (a * a)));
  }
  ETX_DEV void update_height(float in_h) {
    h = in_h;
    C1 = fminf(1.0f, fmaxf(0.0f, 0.5f * (h + 1.0f)));
    if (w.z > 0.9999f)
      G1 = 1.0f;
    else if (w.z <= 0.0f)
      G1 = 0.0f;
    else
      G1 = powf(C1, Lambda);
  }
};
ETX_DEV MsRay ms_ray(const f3& w, const f2 alpha) {
  MsRay r;
  r.Lambda = 0.0f, r.h = 0.0f, r.C1 = 0.0f, r.G1 = 0.0f;
  r.update_direction(w, alpha);
  return r;
}

ETX_DEV float ms_inv_c1(float U) {  // bsdf_external.hxx:72-74
  return fmaxf(-1.0f, fminf(1.0f, 2.0f * U - 1.0f));
}

ETX_DEV float ms_sample_height(const MsRay& ray, float U) {  // bsdf_external.hxx:76-104
  if (ray.w.z > 0.9999f)
    return kMaxFloat;
  if (ray.w.z < -0.9999f)
    return ms_inv_c1(U * ray.C1);
  if (fabsf(ray.w.z) < 0.0001f)
    return ray.h;
  if (U > 1.0f - ray.G1)
    return kMaxFloat;
  float P1 = powf(1.0f - U, 1.0f / ray.Lambda);
  if (P1 <= 0.0f)
    return kMaxFloat;
  return ms_inv_c1(ray.C1 / P1);
}

ETX_DEV float D_ggx(const f3& wm, const f2 alpha) {  // bsdf_external.hxx:106-127
  if (wm.z <= kEpsilon)
    return 0.0f;
  const float slope_x = -wm.x / wm.z, slope_y = -wm.y / wm.z;
  const float ax = fmaxf(kEpsilon, alpha.x * alpha.x);
  const float ay = fmaxf(kEpsilon, alpha.y * alpha.y);
  const float axy = fmaxf(kEpsilon, alpha.x * alpha.y);
  const float tmp = 1.0f + slope_x * slope_x / ax + slope_y * slope_y / ay;
  const float P22 = 1.0f / (kPi * axy * tmp * tmp);
  return P22 / (wm.z * wm.z * wm.z * wm.z);
}

ETX_DEV f2 ms_sample_p22_11(float theta_i, const f2 rnd) {  // bsdf_external.hxx:129-178
  if (theta_i < 0.0001f) {
    const float r = sqrtf(rnd.x / (1.0f - rnd.x));
    const float phi = kDoublePi * rnd.y;
    return {r * cosf(phi), r * sinf(phi)};
  }
  const float sin_theta_i = sinf(theta_i), cos_theta_i = cosf(theta_i);
  const float tan_theta_i = sin_theta_i / cos_theta_i;
  const float projectedarea = 0.5f * (cos_theta_i + 1.0f);
  if (projectedarea < 0.0001f)
    return {0.0f, 0.0f};
  const float c = 1.0f / projectedarea;
  const float A = 2.0f * rnd.x / cos_theta_i / c - 1.0f;
  const float B = tan_theta_i;
  const float tmp = 1.0f / (A * A - 1.0f);
  const float D = sqrtf(fmaxf(0.0f, B * B * tmp * tmp - (A * A - B * B) * tmp));
  const float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
  f2 slope;
  slope.x = (A < 0.0f || slope_x_2 > 1.0f / tan_theta_i) ? slope_x_1 : slope_x_2;
  float U2, S;
  if (rnd.y > 0.5f) {
    S = 1.0f;
    U2 = 2.0f * (rnd.y - 0.5f);
  } else {
    S = -1.0f;
    U2 = 2.0f * (0.5f - rnd.y);
  }
  const float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
  slope.y = S * z * sqrtf(1.0f + slope.x * slope.x);
  return slope;
}

// bsdf_external.hxx:248-279 samplePhaseFunction_conductor
ETX_DEV f3 ms_sample_phase_conductor(const f2 slope_rnd, const f3& wi, const f2 alpha, const Ior& ext_ior, const Ior& int_ior, const ThinfilmEval& tf, f3& weight) {
  const f3 wi_11 = normalize(f3{alpha.x * wi.x, alpha.y * wi.y, wi.z});
  f2 slope_11 = ms_sample_p22_11(acosf(wi_11.z), slope_rnd);
  const float phi = atan2f(wi_11.y, wi_11.x);
  float sp, cp;
  sincosf(phi, &sp, &cp);
  f2 slope = {cp * slope_11.x - sp * slope_11.y, sp * slope_11.x + cp * slope_11.y};
  slope.x *= alpha.x;
  slope.y *= alpha.y;
  f3 wm;
  if ((slope.x != slope.x) || isinf(slope.x)) {
    wm = (wi.z > 0) ? f3{0.0f, 0.0f, 1.0f} : normalize(f3{wi.x, wi.y, 0.0f});
  } else {
    wm = normalize(f3{-slope.x, -slope.y, 1.0f});
  }
  float i_dot_m = dot(wi, wm);
  weight = fresnel_calculate_g(i_dot_m, ext_ior, int_ior, tf);
  return -wi + 2.0f * wm * i_dot_m;
}

// bsdf_external.hxx:211-246 phase_function_reflection
ETX_DEV f3 ms_phase_function_reflection(const MsRay& ray, const f3& wo, const f2 alpha, const Ior& ext_ior, const Ior& int_ior, const ThinfilmEval& tf) {
  if (ray.w.z > 0.9999f)
    return mk3(0.0f);
  float projected_area = (ray.w.z < -0.9999f) ? 1.0f : ray.Lambda * ray.w.z;
  if (projected_area < kEpsilon)
    return mk3(0.0f);
  const f3 wh = normalize(-ray.w + wo);
  if (wh.z < 0.0f)
    return mk3(0.0f);
  float w_dot_h = dot(-ray.w, wh);
  if (w_dot_h < kEpsilon)
    return mk3(0.0f);
  const f3 f = fresnel_calculate_g(w_dot_h, ext_ior, int_ior, tf);
  return f * (D_ggx(wh, alpha) / (4.0f * projected_area));
}

ETX_DEV float ms_mis_weight_conductor(const f3& wi, const f3& wo, const f2 alpha) {  // bsdf_external.hxx:281-287
  if (wi.x == -wo.x && wi.y == -wo.y && wi.z == -wo.z)
    return 1.0f;
  const f3 wh = normalize(wi + wo);
  return D_ggx((wh.z > 0) ? wh : -wh, alpha);
}

// bsdf_external.hxx:289-353 eval_conductor (stochastic: consumes a variable number of randoms)
ETX_DEV f3 ms_eval_conductor(Sampler& smp, const f3& wi, const f3& wo, const f2 alpha, const Ior& ext_ior, const Ior& int_ior, const ThinfilmEval& tf) {
  if (wi.z <= 0 || wo.z <= 0)
    return mk3(0.0f);
  MsRay ray = ms_ray(-wi, alpha);
  ray.update_height(1.0f);
  f3 energy = mk3(1.0f);
  MsRay ray_shadowing = ms_ray(wo, alpha);
  const f3 wh = normalize(wi + wo);
  const float D = D_ggx(wh, alpha);
  const float G2 = 1.0f / (1.0f + (-ray.Lambda - 1.0f) + ray_shadowing.Lambda);
  f3 single_scattering = fresnel_calculate_g(dot(ray.w, wh), ext_ior, int_ior, tf) * (D * G2 / (4.0f * wi.z));
  float wi_mis_weight = 0.0f;
  f3 multiple_scattering = mk3(0.0f);
  uint32_t order = 0;
  while (order < kScatteringOrderMax) {
    ray.update_height(ms_sample_height(ray, smp.next()));
    if (ray.h == kMaxFloat)
      break;
    order++;
    if (order > 1) {
      f3 phase = ms_phase_function_reflection(ray, wo, alpha, ext_ior, int_ior, tf);
      ray_shadowing.update_height(ray.h);
      f3 I = energy * phase * ray_shadowing.G1;
      const float mis = wi_mis_weight / (wi_mis_weight + ms_mis_weight_conductor(-ray.w, wo, alpha));
      multiple_scattering += I * mis;
    }
    f2 slope_rnd = ((order == 1) && smp.has_fixed()) ? f2{smp.fixed_u, smp.fixed_v} : smp.next_2d();
    f3 weight;
    ray.update_direction(ms_sample_phase_conductor(slope_rnd, -ray.w, alpha, ext_ior, int_ior, tf, weight), alpha);
    energy = energy * weight;
    ray.update_height(ray.h);
    if (order == 1)
      wi_mis_weight = ms_mis_weight_conductor(wi, ray.w, alpha);
    if ((ray.h != ray.h) || (ray.w.x != ray.w.x))
      return mk3(0.0f);
  }
  return 0.5f * single_scattering + multiple_scattering;
}

}  // namespace etxd

#include "dev_bsdf_ext.h"

namespace etxd {

// ---------------------------------------------------------------------------------------------------------------
// per-class implementations

// bsdf_various.hxx:36-134 DiffuseBSDF: diffuse_variation 0 Lambert, 1 microfacet random walk, 2 vMF diffuse
ETX_DEV BsdfEval diffuse_layer_v(const DScene& s, const BsdfData& d, const f3& local_w_i, const f3& local_w_o, const etx_abi_material& m, Sampler& smp) {
  if (local_w_o.z <= 0.0f)
    return eval_zero();
  f3 diffuse = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
  BsdfEval e;
  e.eta = 1.0f;
  if (m.diffuse_variation == 1u) {
    e.bsdf = ms_eval_diffuse(smp, local_w_i, local_w_o, evaluate_roughness(s, m, d.tex), diffuse);
    e.func = e.bsdf / local_w_o.z;
  } else if (m.diffuse_variation == 2u) {
    e.func = vmf_diffuse_brdf(local_w_i, local_w_o, evaluate_roughness(s, m, d.tex), diffuse);
    e.bsdf = e.func * local_w_o.z;
  } else {
    e.func = diffuse / kPi;
    e.bsdf = e.func * local_w_o.z;
  }
  e.pdf = kInvPi * local_w_o.z;
  return e;
}

// the Lambert case without a sampler (simple-material kernels, merge-ready camera vertices)
ETX_DEV BsdfEval diffuse_layer(const DScene& s, const BsdfData& d, const f3& local_w_o, const etx_abi_material& m) {
  if (local_w_o.z <= 0.0f)
    return eval_zero();
  f3 diffuse = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
  BsdfEval e;
  e.eta = 1.0f;
  e.func = diffuse / kPi;
  e.bsdf = e.func * local_w_o.z;
  e.pdf = kInvPi * local_w_o.z;
  return e;
}

ETX_DEV BsdfSample diffuse_sample(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {
  Frame frame = normal_frame(d);
  f3 local_w_i = frame.to_local(-d.w_i);
  BsdfSample r = sample_zero();
  r.eta = 1.0f;
  r.properties = kSampleReflection | kSampleDiffuse;
  f3 local_w_o;
  if (m.diffuse_variation == 1u) {
    f3 diffuse = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
    local_w_o = ms_sample_diffuse(smp, local_w_i, evaluate_roughness(s, m, d.tex), diffuse, r.weight);
    r.pdf = kInvPi * local_w_o.z;
  } else {
    f2 cos_rnd = smp.has_fixed() ? f2{smp.fixed_u, smp.fixed_v} : smp.next_2d();
    local_w_o = sample_cosine_distribution(cos_rnd, 1.0f);
    BsdfEval dl = diffuse_layer_v(s, d, local_w_i, local_w_o, m, smp);
    r.weight = (dl.pdf == 0.0f) ? mk3(0.0f) : dl.bsdf / dl.pdf;
    r.pdf = dl.pdf;
  }
  r.w_o = frame.from_local(local_w_o);
  return r;
}

ETX_DEV BsdfEval diffuse_evaluate_v(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m, Sampler& smp) {
  Frame frame = normal_frame(d);
  f3 local_w_o = frame.to_local(w_o);
  if (local_w_o.z <= kEpsilon)
    return eval_zero();
  return diffuse_layer_v(s, d, frame.to_local(-d.w_i), local_w_o, m, smp);
}

ETX_DEV BsdfEval diffuse_evaluate(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m) {
  Frame frame = normal_frame(d);
  f3 local_w_o = frame.to_local(w_o);
  if (local_w_o.z <= kEpsilon)
    return eval_zero();
  return diffuse_layer(s, d, local_w_o, m);
}

ETX_DEV float diffuse_pdf(const BsdfData& d, const f3& w_o) {
  f3 n = dot(d.nrm, d.w_i) < 0.0f ? d.nrm : -d.nrm;  // front_fracing_normal, bsdf.hxx:33-35
  float n_dot_o = dot(n, w_o);
  return (n_dot_o <= kEpsilon) ? 0.0f : kInvPi * n_dot_o;
}

// bsdf_conductor.hxx:13-135 ConductorBSDF
ETX_DEV float conductor_pdf_local(const f3& w_i, const f3& w_o, const f2 roughness) {
  MsRay ray = ms_ray(w_i, roughness);
  // bsdf_conductor.hxx:63,100,123 : the "+ w_o.z" term is the reference's (kept verbatim: it changes MIS weights)
  return D_ggx(normalize(w_o + w_i), roughness) / (1.0f + ray.Lambda) / (4.0f * w_i.z) + w_o.z;
}

ETX_DEV bool conductor_is_delta(const DScene& s, const etx_abi_material& m, const f2 tex) {
  f2 r = evaluate_roughness(s, m, tex);
  return fmaxf(r.x, r.y) <= kDeltaAlphaTreshold;
}

ETX_DEV BsdfSample conductor_sample(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {
  Frame frame = normal_frame(d);
  f3 w_i = frame.to_local(-d.w_i);
  Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  BsdfSample r = sample_zero();
  r.properties = kSampleReflection | (conductor_is_delta(s, m, d.tex) ? kSampleDelta : 0u);
  r.medium_index = d.medium;
  r.eta = 1.0f;
  r.weight = mk3(1.0f);
  f2 roughness = evaluate_roughness(s, m, d.tex);
  MsRay ray = ms_ray(-w_i, roughness);
  ray.update_height(1.0f);
  uint32_t order = 0;
  while (true) {
    ray.update_height(ms_sample_height(ray, smp.next()));
    if (ray.h == kMaxFloat)
      break;
    f2 slope_rnd = ((order == 0) && smp.has_fixed()) ? f2{smp.fixed_u, smp.fixed_v} : smp.next_2d();
    f3 weight;
    ray.update_direction(ms_sample_phase_conductor(slope_rnd, -ray.w, roughness, ext_ior, int_ior, tf, weight), roughness);
    ray.update_height(ray.h);
    r.weight *= weight;
    if ((order++ > kScatteringOrderMax) || (ray.h != ray.h) || (ray.w.x != ray.w.x)) {
      r.weight = mk3(0.0f);
      ray.w = f3{0.0f, 0.0f, 1.0f};
      break;
    }
  }
  f3 local_w_o = ray.w;
  r.weight *= apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
  r.pdf = conductor_pdf_local(w_i, local_w_o, roughness);
  r.w_o = normalize(frame.from_local(local_w_o));
  return r;
}

ETX_DEV BsdfEval conductor_evaluate(const DScene& s, const BsdfData& d, const f3& in_w_o, const etx_abi_material& m, Sampler& smp) {
  Frame frame = normal_frame(d);
  f3 w_o = frame.to_local(in_w_o);
  if (w_o.z <= kEpsilon)
    return eval_zero();
  f3 w_i = frame.to_local(-d.w_i);
  if (w_i.z <= kEpsilon)
    return eval_zero();
  f2 roughness = evaluate_roughness(s, m, d.tex);
  Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  const ThinfilmEval tf = evaluate_thinfilm(s, m.thinfilm, d.tex, smp, d.wavelength);
  f3 value = ms_eval_conductor(smp, w_i, w_o, roughness, ext_ior, int_ior, tf);
  BsdfEval e;
  e.eta = 1.0f;
  e.bsdf = value * apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
  e.func = e.bsdf / w_o.z;
  e.pdf = conductor_pdf_local(w_i, w_o, roughness);
  return e;
}

ETX_DEV float conductor_pdf(const DScene& s, const BsdfData& d, const f3& in_w_o, const etx_abi_material& m) {
  Frame frame = normal_frame(d);
  f3 w_o = frame.to_local(in_w_o);
  if (w_o.z <= kEpsilon)
    return 0.0f;
  f3 w_i = frame.to_local(-d.w_i);
  if (w_i.z <= kEpsilon)
    return 0.0f;
  return conductor_pdf_local(w_i, w_o, evaluate_roughness(s, m, d.tex));
}

// bsdf_various.hxx:213-256 MirrorBSDF
ETX_DEV bool direction_matches(const f3& ideal, const f3& actual) {  // math.hxx:1087-1091
  return dot(normalize(ideal), normalize(actual)) > 1.0f - kInvMaxHalf;
}

// bsdf_various.hxx:136-211 TranslucentBSDF
ETX_DEV BsdfSample translucent_sample(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {
  Frame frame = normal_frame(d);
  f3 tr = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
  f3 rf = apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
  float tr_value = luminance(tr), rf_value = luminance(rf);
  float total = tr_value + rf_value;
  if (total == 0.0f)
    return sample_zero();
  f3 w_o = sample_cosine_distribution(smp.next_2d(), frame.nrm, 1.0f);
  float n_dot_o = fabsf(dot(w_o, frame.nrm));
  BsdfSample r = sample_zero();
  r.eta = 1.0f;
  if (smp.next() < tr_value / total) {
    r.w_o = -w_o;
    r.pdf = n_dot_o * kInvPi * (tr_value / total);
    r.properties = kSampleDiffuse | kSampleTransmission | kSampleMediumChanged;
    r.medium_index = frame.entering ? m.int_medium : m.ext_medium;
    r.weight = tr;
  } else {
    r.w_o = w_o;
    r.pdf = n_dot_o * kInvPi * (rf_value / total);
    r.properties = kSampleDiffuse | kSampleReflection;
    r.medium_index = d.medium;
    r.weight = rf;
  }
  return r;
}

ETX_DEV BsdfEval translucent_evaluate(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m) {
  Frame frame = normal_frame(d);
  float n_dot_i = -dot(frame.nrm, d.w_i);
  float n_dot_o = dot(frame.nrm, w_o);
  bool reflection = n_dot_o * n_dot_i > 0.0f;
  f3 tr = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
  f3 rf = apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
  float tr_value = luminance(tr), rf_value = luminance(rf);
  float total = tr_value + rf_value;
  if (total == 0.0f)
    return eval_zero();
  float scale = (total > 1.0f) ? 1.0f / total : 1.0f;
  n_dot_o = fabsf(n_dot_o);
  BsdfEval e;
  e.eta = 1.0f;
  e.func = (reflection ? rf : tr) * (scale * kInvPi);
  e.bsdf = e.func * n_dot_o;
  e.pdf = kInvPi * n_dot_o * (reflection ? rf_value / total : tr_value / total);
  return e;
}

ETX_DEV float translucent_pdf(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m) {
  Frame frame = normal_frame(d);
  float n_dot_i = -dot(frame.nrm, d.w_i);
  float n_dot_o = dot(frame.nrm, w_o);
  float tr_value = luminance(apply_image(s, m.scattering, d.tex, nullptr, d.wavelength));
  float rf_value = luminance(apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength));
  float total = tr_value + rf_value;
  bool reflection = n_dot_o * n_dot_i > 0.0f;
  return (total == 0.0f) ? 0.0f : kInvPi * fabsf(n_dot_o) * (reflection ? rf_value / total : tr_value / total);
}

// ---------------------------------------------------------------------------------------------------------------
// dispatch  scene_bsdf.hxx:56-107

// Mirror (bsdf_various.hxx:215-224) and Boundary (:260-270)
ETX_DEV BsdfSample bsdf_sample_delta_classes(const DScene& s, const BsdfData& d, const etx_abi_material& m) {
  BsdfSample r = sample_zero();
  if (m.cls == ETX_MAT_MIRROR) {
    Frame frame = normal_frame(d);
    r.w_o = normalize(reflect(d.w_i, frame.nrm));
    r.weight = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
    r.pdf = 1.0f;
    r.properties = kSampleDelta | kSampleReflection;
    return r;
  }
  bool entering = dot(d.nrm, d.w_i) < 0.0f;
  r.w_o = d.w_i;
  r.pdf = 1.0f;
  r.weight = mk3(1.0f);
  r.properties = kSampleTransmission | kSampleMediumChanged;
  r.medium_index = entering ? m.int_medium : m.ext_medium;
  return r;
}

enum : uint32_t { kPrincipledConductor = 0, kPrincipledDielectric = 1, kPrincipledPlastic = 2 };  // material variants (dev_bsdf_ool.h)

// ---------------------------------------------------------------------------------------------------------------
// "Simple material" dispatch. Surfaces whose only non-diffuse lobes are perfect mirrors (Mirror, Conductor with
// roughness exactly 0) - every Cornell wall - never need the Heitz random walk: with alpha = 0 the walk of
// ConductorBSDF::sample (bsdf_conductor.hxx:13-70) deterministically reflects about the shading normal after one
// microsurface interaction (height sample leaves the surface with probability 1, bsdf_external.hxx:76-104), the weight
// is the Fresnel term of the macro normal and the "pdf" is the same D_ggx expression. Paths that hit such a material
// are shaded by the simple kernels (2 -> 4 waves per SIMD), the others are binned into the general kernels
// (DScene::material_group, kernels_shade.inl); the general dispatch is dev_bsdf_ool.h.
ETX_DEV BsdfSample conductor_sample_delta(const DScene& s, const BsdfData& d, const etx_abi_material& m) {
  Frame frame = normal_frame(d);
  f3 w_i = frame.to_local(-d.w_i);
  Ior ext_ior = evaluate_refractive_index(s, m.ext_ior, d.wavelength);
  Ior int_ior = evaluate_refractive_index(s, m.int_ior, d.wavelength);
  BsdfSample r = sample_zero();
  r.properties = kSampleReflection | kSampleDelta;
  r.medium_index = d.medium;
  r.eta = 1.0f;
  f3 local_w_o = {-w_i.x, -w_i.y, w_i.z};  // -wi + 2 wm (wi . wm) with wm = (0, 0, 1)
  r.weight = fresnel_calculate(w_i.z, ext_ior, int_ior, thinfilm_none()) * apply_image(s, m.reflectance, d.tex, nullptr, d.wavelength);
  r.pdf = conductor_pdf_local(w_i, local_w_o, f2{0.0f, 0.0f});
  r.w_o = normalize(frame.from_local(local_w_o));
  return r;
}

// Lambert DiffuseBSDF::sample without the rough-diffuse variations (bsdf_various.hxx:88-110, default branch)
ETX_DEV BsdfSample diffuse_sample_lambert(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {
  Frame frame = normal_frame(d);
  BsdfSample r = sample_zero();
  r.eta = 1.0f;
  r.properties = kSampleReflection | kSampleDiffuse;
  f2 cos_rnd = smp.has_fixed() ? f2{smp.fixed_u, smp.fixed_v} : smp.next_2d();
  f3 local_w_o = sample_cosine_distribution(cos_rnd, 1.0f);
  BsdfEval dl = diffuse_layer(s, d, local_w_o, m);
  r.weight = (dl.pdf == 0.0f) ? mk3(0.0f) : dl.bsdf / dl.pdf;
  r.pdf = dl.pdf;
  r.w_o = frame.from_local(local_w_o);
  return r;
}

ETX_DEV BsdfSample bsdf_sample_simple(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {
  switch (m.cls) {
    case ETX_MAT_DIFFUSE:
      return diffuse_sample_lambert(s, d, m, smp);
    case ETX_MAT_TRANSLUCENT:
      return translucent_sample(s, d, m, smp);
    case ETX_MAT_CONDUCTOR:
      return conductor_sample_delta(s, d, m);
    case ETX_MAT_MIRROR:
    case ETX_MAT_BOUNDARY:
      return bsdf_sample_delta_classes(s, d, m);
    default: {  // Void
      BsdfSample r = sample_zero();
      r.w_o = d.w_i;
      r.properties = kSampleDelta;
      r.medium_index = d.medium;
      return r;
    }
  }
}
ETX_DEV BsdfEval bsdf_evaluate_simple(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m) {
  switch (m.cls) {  // delta conductors are never evaluated (only connectible vertices are)
    case ETX_MAT_DIFFUSE:
      return diffuse_evaluate(s, d, w_o, m);
    case ETX_MAT_TRANSLUCENT:
      return translucent_evaluate(s, d, w_o, m);
    default:
      return eval_zero();
  }
}
ETX_DEV float bsdf_pdf_simple(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m) {
  switch (m.cls) {
    case ETX_MAT_DIFFUSE:
      return diffuse_pdf(d, w_o);
    case ETX_MAT_TRANSLUCENT:
      return translucent_pdf(s, d, w_o, m);
    default:
      return 0.0f;
  }
}

// Lambert surfaces take the sampler-free fast paths of the connect / merge kernels
ETX_HD bool material_is_lambert(const etx_abi_material& m) {
  return (m.cls == ETX_MAT_DIFFUSE) && (m.diffuse_variation == 0u);
}

// classes the device path implements (checked by the host at upload)
ETX_HD bool bsdf_class_supported(uint32_t cls) {
  return cls < ETX_MAT_COUNT;  // all eleven classes of scene_bsdf.hxx:56-107
}

}  // namespace etxd
