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
// Random-walk subsurface scattering: subsurface::remap_channel (scene_bssrdf_subsurface.hxx:17-44) and
// subsurface::gather_rw (path_tracing_shared.hxx:64-159). The walk runs inside the shade kernel with an inline
// material-filtered closest-hit query (Raytracing::trace_material, rt.cxx:327-371) until it leaves the object.
ETX_DEV void sss_remap_channel(float color, float scattering_distance, float& albedo, float& extinction, float& scattering) {
  const float a = 1.826052378200f, b = 4.985111943850f + 0.12735595943800f, c = 1.096861024240f;
  const float d = 0.496310210422f, e = 4.231902997010f + 0.00310603949088f, f = 2.406029994080f;
  const float kMinScattering = 1.0f / 1024.0f;
  color = fmaxf(0.0f, color);
  const float blend = powf(color, 0.25f);
  albedo = (1.0f - blend) * a * powf(atanf(b * color), c) + blend * d * powf(atanf(e * color), f);
  albedo = fminf(fmaxf(albedo, 0.0f), 1.0f - kEpsilon);
  extinction = 1.0f / fmaxf(scattering_distance, kMinScattering);
  scattering = extinction * albedo;
}

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

}  // namespace etxd
