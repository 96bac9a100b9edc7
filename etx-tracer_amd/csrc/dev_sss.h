// dev_sss.h - random-walk subsurface scattering shared by the PT and VCM step functions.
#pragma once

#include "dev_bvh.h"
#include "dev_bsdf_ool.h"
#include "dev_emitters.h"

namespace etxd {

ETX_DEV bool is_zero_sss(const f3& v) {
  return (v.x <= kEpsilon) && (v.y <= kEpsilon) && (v.z <= kEpsilon);
}

// ---------------------------------------------------------------------------------------------------------------
// Random-walk subsurface scattering: subsurface::remap_channel (scene_bssrdf_subsurface.hxx:17-44, dev_scene.h sss_remap_channel) and
// subsurface::gather_rw (path_tracing_shared.hxx:64-159). The walk runs inside the shade kernel with an inline
// material-filtered closest-hit query (Raytracing::trace_material, rt.cxx:327-371) until it leaves the object.
ETX_DEV float sss_safe_mul(float a, float b) {  // path_tracing_shared.hxx:51-53
  return ((a == 0.0f) || (b == 0.0f)) ? 0.0f : a * b;
}

ETX_DEV bool sss_gather_rw(const DScene& scene, const LaneStack& stack, const Isect& in, Sampler& smp, float wavelength, Isect& out, f3& out_weight) {
  const uint32_t kMaxIterations = 1024u;
  const etx_abi_material& mat = scene.materials[in.material];
  float anisotropy = 0.0f;
  f3 extinction, scattering, albedo;
  if (mat.int_medium == kInvalid) {
    const f3 color = apply_image(scene, mat.scattering, in.tex, nullptr, wavelength);
    const etx_abi_spectral_image distances_image = {mat.subsurface.spectrum_index, mat.subsurface.image_index};
    const f3 distances = apply_image(scene, distances_image, in.tex, nullptr, wavelength);
    sss_remap_channel(color.x, distances.x, albedo.x, extinction.x, scattering.x);
    sss_remap_channel(color.y, distances.y, albedo.y, extinction.y, scattering.y);
    sss_remap_channel(color.z, distances.z, albedo.z, extinction.z, scattering.z);
  } else {
    const DMedium& medium = scene.mediums[mat.int_medium];
    anisotropy = medium.g;
    f3 absorption;
    medium_coefficients(scene, medium, wavelength, absorption, scattering);
    extinction = scattering + absorption;
    albedo = {extinction.x > 0.0f ? scattering.x / extinction.x : 0.0f, extinction.y > 0.0f ? scattering.y / extinction.y : 0.0f, extinction.z > 0.0f ? scattering.z / extinction.z : 0.0f};
  }
  f3 ray_d = (mat.subsurface.path == 0u) ? sample_cosine_distribution(smp.next_2d(), -in.nrm, 1.0f) : in.w_i;  // Path::Diffuse
  f3 ray_o = shading_pos(scene, scene.triangles[in.tri], in.bc, ray_d);
  f3 throughput = mk3(1.0f);
  uint32_t alpha_seed = smp.seed ^ 0x73737321u;
  for (uint32_t i = 0; i < kMaxIterations; ++i) {
    f3 pdf;
    const uint32_t channel = sample_spectrum_component(albedo, throughput, smp.next(), pdf);
    const float scattering_distance = channel == 0 ? extinction.x : (channel == 1 ? extinction.y : extinction.z);
    float max_t = scattering_distance > 0.0f ? (-logf(1.0f - smp.next()) / scattering_distance) : kMaxFloat;
    if ((i == 0u) && (max_t <= kRayEpsilon))
      return false;
    const Hit h = bvh_closest(scene, global_nodes(scene), scene.bvh_tris, scene.bvh_root, stack, RayQ{ray_o, kRayEpsilon, ray_d, max_t}, alpha_seed, nullptr, in.material);
    const bool found = h.tri != kInvalid;
    if (found)
      max_t = h.t;
    const f3 tr = {expf(-max_t * extinction.x), expf(-max_t * extinction.y), expf(-max_t * extinction.z)};
    pdf = found ? pdf * tr : pdf * f3{sss_safe_mul(tr.x, extinction.x), sss_safe_mul(tr.y, extinction.y), sss_safe_mul(tr.z, extinction.z)};
    if (is_zero_sss(pdf))
      return false;
    const f3 weight = found ? tr : f3{sss_safe_mul(tr.x, scattering.x), sss_safe_mul(tr.y, scattering.y), sss_safe_mul(tr.z, scattering.z)};
    throughput *= weight / (pdf.x + pdf.y + pdf.z);
    if (max_component(throughput) <= kEpsilon)
      return false;
    if (found) {
      out = make_intersection(scene, ray_d, h.u, h.v, h.t, h.tri);
      const bool w_i_in = dot(out.w_i, out.nrm) > 0.0f;
      out.w_i = out.w_i * (w_i_in ? -1.0f : 1.0f);
      out_weight = throughput;
      return true;
    }
    const f3 prev_dir = ray_d;
    ray_o = ray_o + ray_d * max_t;
    ray_d = sample_phase_function(prev_dir, anisotropy, smp.next_2d());
  }
  return false;
}

// ---------------------------------------------------------------------------------------------------------------
// Christensen-Burley subsurface scattering: subsurface::sample / evaluate / geometric_weigth
// (scene_bssrdf_subsurface.hxx:46-146), subsurface::gather_cb (path_tracing_shared.hxx:149-220) and
// Raytracing::continuous_trace (rt.cxx:373-426).
//
// The reference shoots three probe rays (one per axis of the shading frame), collects up to eight hits of the object's
// material on each (continuous_trace), weighs every hit, lets the caller connect / light EVERY hit with its weight, and
// continues from one of them chosen in proportion to the weights. Nothing is buffered here: the hits of a probe ray are
// found one after the other with the material-filtered closest-hit query (the next query starts behind the last hit), each
// is handed to `emit(exit point, weight)` at once, and the continuation point is picked by weighted reservoir sampling -
// the same distribution as the reference's `rnd * total_weight` over the buffered list.
ETX_DEV f3 sss_cb_evaluate(const DScene& scene, const etx_abi_material& mat, const f2& tex, float wavelength, float radius) {  // scene_bssrdf_subsurface.hxx:57-75
  const etx_abi_spectral_image distances_image = {mat.subsurface.spectrum_index, mat.subsurface.image_index};
  const f3 sd = apply_image(scene, distances_image, tex, nullptr, wavelength);
  radius = fmaxf(radius, kEpsilon);
  const f3 term_0 = {expf(-radius / (3.0f * sd.x)), expf(-radius / (3.0f * sd.y)), expf(-radius / (3.0f * sd.z))};
  const f3 term_1 = term_0 * term_0 * term_0;
  const float k = 4.0f * radius * kDoublePi;
  const f3 div = {fmaxf(sd.x * k, kEpsilon), fmaxf(sd.y * k, kEpsilon), fmaxf(sd.z * k, kEpsilon)};
  return (term_0 + term_1) / div;
}

ETX_DEV float sss_cb_sample_s_r(float rnd) {  // :46-55
  if (rnd < 0.25f) {
    rnd = fminf(4.0f * rnd, 1.0f - kEpsilon);
    return logf(1.0f / (1.0f - rnd));
  }
  rnd = fminf((rnd - 0.25f) / 0.75f, 1.0f - kEpsilon);
  return 3.0f * logf(1.0f / (1.0f - rnd));
}

struct SssProbe {  // subsurface::Sample, :77-88
  f3 ray_o, ray_d, u, v, w, basis_prob;
  float max_t, sampled_radius;
  bool valid;
};

ETX_DEV SssProbe sss_cb_sample(const DScene& scene, const Isect& in, const etx_abi_material& mat, uint32_t direction, Sampler& smp, float wavelength) {  // :90-138
  SssProbe r = {};
  r.valid = false;
  const etx_abi_spectral_image distances_image = {mat.subsurface.spectrum_index, mat.subsurface.image_index};
  const f3 sampled_distance = apply_image(scene, distances_image, in.tex, nullptr, wavelength);
  // SpectralResponse::component_count: 3 in RGB mode, 1 in spectral mode (the three components are replicas here)
  const uint32_t channel = scene.spectral ? 0u : min(2u, uint32_t(3.0f * smp.next()));
  if (scene.spectral)
    (void)smp.next();
  const float scattering_distance = (channel == 0u) ? sampled_distance.x : ((channel == 1u) ? sampled_distance.y : sampled_distance.z);
  if (scattering_distance == 0.0f)
    return r;
  if (direction == 0u)
    r.u = in.tan, r.v = in.btn, r.w = in.nrm, r.basis_prob = {0.25f, 0.25f, 0.5f};
  else if (direction == 1u)
    r.u = in.btn, r.v = in.nrm, r.w = in.tan, r.basis_prob = {0.25f, 0.50f, 0.25f};
  else
    r.u = in.nrm, r.v = in.tan, r.w = in.btn, r.basis_prob = {0.5f, 0.25f, 0.25f};
  const float kMaxRadius = 47.827155457397595950044717258511f;
  const float r_max = scattering_distance * kMaxRadius;
  r.sampled_radius = scattering_distance * sss_cb_sample_s_r(smp.next());
  if (r.sampled_radius >= r_max)
    return r;
  float sn, cs;
  sincos_rev(smp.next(), &sn, &cs);
  const float height = sqrtf(sqr(r_max) - sqr(r.sampled_radius));
  if (height <= kRayEpsilon)
    return r;
  r.ray_o = in.pos + r.w * height + (r.u * cs + r.v * sn) * r.sampled_radius;
  r.ray_d = -r.w;
  r.max_t = 2.0f * height;
  r.valid = true;
  return r;
}

ETX_DEV float sss_cb_geometric_weight(const f3& nrm, const SssProbe& s) {  // :140-145
  const float pdf_t = s.basis_prob.x * fabsf(dot(nrm, s.u));
  const float pdf_b = s.basis_prob.y * fabsf(dot(nrm, s.v));
  const float pdf_n = s.basis_prob.z * fabsf(dot(nrm, s.w));
  return sqr(pdf_n) / (sqr(pdf_t) + sqr(pdf_b) + sqr(pdf_n));
}

// Returns false when nothing was gathered (the path ends, path_tracing_shared.hxx:401-403). `selected` / `selected_weight`:
// the exit point the path continues from and weights[selected] * selected_sample_weight (:438-440).
template <class Emit>
ETX_DEV bool sss_gather_cb(const DScene& scene, const LaneStack& stack, const Isect& in, Sampler& smp, float wavelength, Isect& selected, f3& selected_weight, Emit&& emit) {
  constexpr uint32_t kIntersectionsPerDirection = 8u;  // scene_bssrdf_subsurface.hxx:5
  const etx_abi_material& mat = scene.materials[in.material];
  // the three probes are sampled before the first trace, like the reference's initialiser list (:153-157)
  SssProbe probes[3] = {sss_cb_sample(scene, in, mat, 0u, smp, wavelength), sss_cb_sample(scene, in, mat, 1u, smp, wavelength), sss_cb_sample(scene, in, mat, 2u, smp, wavelength)};
  const f3 base_weight = apply_image(scene, mat.scattering, in.tex, nullptr, wavelength);
  uint32_t alpha_seed = smp.seed ^ 0x63627373u;
  float total_weight = 0.0f, selected_average = 0.0f;
  f3 selected_raw = mk3(0.0f);
  uint32_t count = 0u;
#pragma unroll 1
  for (uint32_t d = 0; d < 3u; ++d) {
    const SssProbe& probe = probes[d];
    if (probe.valid == false)  // a default Sample has a zero-length ray: continuous_trace finds nothing
      continue;
    float t_min = kRayEpsilon;
#pragma unroll 1
    for (uint32_t k = 0; k < kIntersectionsPerDirection; ++k) {  // continuous_trace: every hit of this material along the ray, at most eight
      const Hit h = bvh_closest(scene, global_nodes(scene), scene.bvh_tris, scene.bvh_root, stack, RayQ{probe.ray_o, t_min, probe.ray_d, probe.max_t}, alpha_seed, nullptr, in.material);
      if (h.tri == kInvalid)
        break;
      t_min = h.t + fmaxf(kRayEpsilon, h.t * 1.0e-6f);
      const Isect out = make_intersection(scene, probe.ray_d, h.u, h.v, h.t, h.tri);
      const float gw = sss_cb_geometric_weight(out.nrm, probe);
      const f3 pdf3 = sss_cb_evaluate(scene, mat, out.tex, wavelength, probe.sampled_radius);
      const float pdf = scene.spectral ? pdf3.x : (pdf3.x + pdf3.y + pdf3.z) / 3.0f;  // SpectralResponse::average
      if ((pdf > 0.0f) == false)
        continue;
      const f3 eval = sss_cb_evaluate(scene, mat, out.tex, wavelength, length(out.pos - in.pos));
      const f3 weight = base_weight * eval * (gw / pdf);
      if ((weight.x == 0.0f) && (weight.y == 0.0f) && (weight.z == 0.0f))
        continue;
      const float average = scene.spectral ? weight.x : (weight.x + weight.y + weight.z) / 3.0f;
      total_weight += average;
      count += 1u;
      emit(out, weight);
      if (smp.next() * total_weight < average) {  // weighted reservoir: P(selected = i) = average_i / total
        selected = out;
        selected_raw = weight;
        selected_average = average;
      }
    }
  }
  if ((count == 0u) || ((total_weight > 0.0f) == false) || (selected_average == 0.0f))
    return false;
  selected_weight = selected_raw * (total_weight / selected_average);
  return true;
}

}  // namespace etxd
