// dev_emitters.h - emitters, camera and participating media (RGB mode).
// Follows sources/etx/render/shared/scene_emitters.hxx, scene_camera.hxx, scene_medium.hxx (cited per function).
#pragma once

#include "dev_scene.h"

namespace etxd {

// ---------------------------------------------------------------------------------------------------------------
// emitters

struct EmitterSample {  // emitter.hxx:73-96
  f3 value;
  f3 barycentric;
  float pdf_sample;
  f3 origin;
  float pdf_area;
  f3 normal;
  float pdf_dir;
  f3 direction;
  float pdf_dir_out;
  uint32_t emitter_index, triangle_index, medium_index;
  bool is_delta, is_distant;
};

ETX_DEV EmitterSample emitter_sample_zero() {
  EmitterSample r;
  r.value = r.barycentric = r.origin = r.normal = r.direction = mk3(0.0f);
  r.pdf_sample = r.pdf_area = r.pdf_dir = r.pdf_dir_out = 0.0f;
  r.emitter_index = r.triangle_index = r.medium_index = kInvalid;
  r.is_delta = r.is_distant = false;
  return r;
}

struct EmitterRadianceQuery {  // emitter.hxx:98-104
  f3 source_position, target_position, direction;
  f2 uv;
  bool directly_visible;
};

ETX_DEV float collimation_to_exponent(float normalized) {  // scene.hxx:67-71
  float t = saturate(normalized);
  float denom = sqr(sqr(1.0f - t));
  return 1.0f / fmaxf(kEpsilon, denom);
}

// The emitter records (32 bytes) and their profiles (48 bytes, 16-byte aligned) sit in tables the upload aligns: read as whole 16-byte rows - two and
// three loads instead of a dword gather per field (a gather costs the same whatever its width) - and handed on by value.
ETX_DEV etx_abi_emitter load_emitter(const DScene& s, uint32_t index) {
  const float4* r = reinterpret_cast<const float4*>(s.emitters + index);
  const float4 a = r[0], b = r[1];
  etx_abi_emitter e;
  e.cls = __float_as_uint(a.x), e.profile = __float_as_uint(a.y), e.triangle_index = __float_as_uint(a.z);
  e.spectrum_weight = a.w, e.additional_weight = b.x, e.triangle_area = b.y, e.pad0 = 0.0f, e.pad1 = 0.0f;
  return e;
}
ETX_DEV etx_abi_emitter_profile load_emitter_profile(const DScene& s, uint32_t index) {
  const float4* r = reinterpret_cast<const float4*>(s.emitter_profiles + index);
  const float4 a = r[0], b = r[1], c = r[2];
  etx_abi_emitter_profile p;
  p.emission.spectrum_index = __float_as_uint(a.x), p.emission.image_index = __float_as_uint(a.y);
  p.direction = {a.z, a.w, b.x};
  p.cls = __float_as_uint(b.y);
  p.angular_size = b.z, p.equivalent_disk_size = b.w, p.angular_size_cosine = c.x, p.pad0 = 0.0f, p.pad1 = 0.0f;
  return p;
}
static_assert((sizeof(etx_abi_emitter) == 32) && (sizeof(etx_abi_emitter_profile) == 48), "row loads of the emitter tables");

ETX_DEV uint32_t emitter_external_medium_index(const DScene& s, const etx_abi_emitter& em) {  // scene_emitters.hxx:10-19
  if ((em.cls != ETX_EMITTER_AREA) || (em.triangle_index >= s.triangle_count))
    return kInvalid;
  const uint32_t material_index = __float_as_uint(s.tri_shade[size_t(em.triangle_index) * kTriShadeStride + 6u].w);  // row 6: geometric normal, material
  if (material_index >= s.material_count)
    return kInvalid;
  return s.materials[material_index].ext_medium;
}

ETX_DEV float emitter_discrete_pdf(const DScene& s, const etx_abi_emitter& em) {  // scene_emitters.hxx:205-207
  return (em.spectrum_weight * em.additional_weight) / s.emitter_dist_total;
}

ETX_DEV float env_pdf_area(const DScene& s) {
  return 1.0f / (kPi * s.bounds_radius * s.bounds_radius);
}

// scene_emitters.hxx:40-105 emitter_get_radiance
ETX_DEV f3 emitter_get_radiance(const DScene& s, const etx_abi_emitter& em_inst, const etx_abi_emitter_profile& em, const EmitterRadianceQuery& q, float& pdf_area, float& pdf_dir,
  float& pdf_dir_out, float wavelength) {
  pdf_dir = 0.0f, pdf_area = 0.0f, pdf_dir_out = 0.0f;
  switch (em_inst.cls) {
    case ETX_EMITTER_DIRECTIONAL: {
      f3 em_dir = ld3(em.direction);
      if ((q.directly_visible == false) || (em.angular_size <= 0.0f) || (dot(q.direction, em_dir) < em.angular_size_cosine))
        return mk3(0.0f);
      pdf_dir = 1.0f;
      pdf_area = env_pdf_area(s);
      pdf_dir_out = pdf_dir * pdf_area;
      f2 uv = disk_uv(em_dir, q.direction, em.equivalent_disk_size, em.angular_size_cosine);
      f3 direct_scale = mk3(1.0f) / (spectrum_eval(s, em.emission.spectrum_index, wavelength) * (kDoublePi * (1.0f - em.angular_size_cosine)));
      return apply_image(s, em.emission, uv, nullptr, wavelength) * direct_scale;
    }
    case ETX_EMITTER_ENVIRONMENT: {
      const DImage& img = s.images[em.emission.image_index];
      f2 uv = direction_to_uv(q.direction, img.offset, img.scale.x);
      float sin_t = fmaxf(kEpsilon, sin_rev(uv.y * 0.5f));
      float image_pdf = 0.0f;
      f3 eval = apply_image(s, em.emission, uv, &image_pdf, wavelength);
      pdf_area = env_pdf_area(s);
      pdf_dir = image_pdf / (2.0f * kPi * kPi * sin_t);
      pdf_dir_out = pdf_area * pdf_dir;
      return eval;
    }
    default: {  // Area
      const float4 row = s.tri_shade[size_t(em_inst.triangle_index) * kTriShadeStride + 6u];  // geometric normal, material index
      const etx_abi_material& material = s.materials[__float_as_uint(row.w)];
      f3 geo_n = xyz(row);
      if (dot(geo_n, q.target_position - q.source_position) >= 0.0f)
        return mk3(0.0f);
      pdf_area = 1.0f / em_inst.triangle_area;
      f3 dp = q.source_position - q.target_position;
      float distance_squared = dot(dp, dp);
      if (distance_squared > 0.0f) {
        float cos_t = fabsf(dot(dp, geo_n)) / sqrtf(distance_squared);
        float exponent = collimation_to_exponent(material.emission_collimation);
        float cos_tx = (q.directly_visible || (exponent == 1.0f)) ? cos_t : powf(cos_t, exponent);
        if (cos_tx > kEpsilon) {
          pdf_dir = pdf_area * distance_squared / cos_tx;
          pdf_dir_out = pdf_area * cos_tx * kInvPi;
        }
      }
      return apply_image(s, em.emission, q.uv, nullptr, wavelength);
    }
  }
}

ETX_DEV f3 emitter_get_radiance(const DScene& s, const etx_abi_emitter& em_inst, const EmitterRadianceQuery& q, float& pdf_area, float& pdf_dir, float& pdf_dir_out, float wavelength) {
  return emitter_get_radiance(s, em_inst, load_emitter_profile(s, em_inst.profile), q, pdf_area, pdf_dir, pdf_dir_out, wavelength);
}

// scene_emitters.hxx:139-203 emitter_sample_in + :216-224 sample_emitter
ETX_DEV EmitterSample sample_emitter(const DScene& s, uint32_t emitter_index, const f2 smp, const f3& from_point, float wavelength) {
  const etx_abi_emitter em_inst = load_emitter(s, emitter_index);
  const etx_abi_emitter_profile em = load_emitter_profile(s, em_inst.profile);
  EmitterSample r = emitter_sample_zero();
  switch (em_inst.cls) {
    case ETX_EMITTER_AREA: {
      const etx_abi_triangle& tri = s.triangles[em_inst.triangle_index];
      r.barycentric = random_barycentric(smp);
      EmitterRadianceQuery q;
      lerp_pos_normal_uv(s, tri, r.barycentric, r.origin, r.normal, q.uv);  // lerp_pos, lerp_normal, lerp_uv from the six rows they share
      r.direction = normalize(r.origin - from_point);
      q.source_position = from_point;
      q.target_position = r.origin;
      q.direction = mk3(0.0f);
      q.directly_visible = false;
      r.value = emitter_get_radiance(s, em_inst, em, q, r.pdf_area, r.pdf_dir, r.pdf_dir_out, wavelength);
      break;
    }
    case ETX_EMITTER_DIRECTIONAL: {
      f3 em_dir = ld3(em.direction);
      f2 disk_sample = {0.0f, 0.0f};
      if (em.angular_size > 0.0f) {
        Basis basis = orthonormal_basis(em_dir);
        disk_sample = sample_disk(smp);
        r.direction = normalize(em_dir + basis.u * disk_sample.x * (0.5f * em.equivalent_disk_size) + basis.v * disk_sample.y * (0.5f * em.equivalent_disk_size));
      } else {
        r.direction = em_dir;
      }
      r.pdf_area = env_pdf_area(s);
      r.pdf_dir = 1.0f;
      r.pdf_dir_out = r.pdf_dir * r.pdf_area;
      r.origin = from_point + r.direction * distance_to_sphere(from_point, r.direction, s.bounds_center, s.bounds_radius);
      r.normal = em_dir * (-1.0f);
      r.value = apply_image(s, em.emission, disk_sample * 0.5f + f2{0.5f, 0.5f}, nullptr, wavelength);
      break;
    }
    default: {  // Environment
      const DImage& img = s.images[em.emission.image_index];
      float pdf_image = 0.0f;
      float4 image_value = make_float4(0, 0, 0, 0);
      f2 uv = image_sample(img, smp, pdf_image, image_value);
      float sin_t = fmaxf(kEpsilon, sin_rev(uv.y * 0.5f));
      r.direction = uv_to_direction(uv, img.offset, img.scale.x);
      r.normal = -r.direction;
      r.origin = from_point + r.direction * distance_to_sphere(from_point, r.direction, s.bounds_center, s.bounds_radius);
      r.pdf_dir = pdf_image / (2.0f * kPi * kPi * sin_t);
      r.pdf_area = env_pdf_area(s);
      r.pdf_dir_out = r.pdf_area * r.pdf_dir;
      r.value = spectrum_eval(s, em.emission.spectrum_index, wavelength) * mk3(image_value);
      break;
    }
  }
  r.medium_index = emitter_external_medium_index(s, em_inst);
  r.pdf_sample = emitter_discrete_pdf(s, em_inst);
  r.emitter_index = emitter_index;
  r.triangle_index = em_inst.triangle_index;
  r.is_delta = em_inst.cls == ETX_EMITTER_DIRECTIONAL;
  return r;
}

ETX_DEV uint32_t sample_emitter_index(const DScene& s, float rnd) {  // scene_emitters.hxx:209-214
  return distribution_sample(s.emitter_dist, s.emitter_dist_count, rnd);
}

// scene_emitters.hxx:226-306 sample_emission : start of a light sub path
ETX_DEV EmitterSample sample_emission(const DScene& s, Sampler& smp, float wavelength) {
  EmitterSample r = emitter_sample_zero();
  r.emitter_index = distribution_sample(s.emitter_dist, s.emitter_dist_count, smp.next());
  r.pdf_sample = s.emitter_dist[r.emitter_index].pdf;
  const etx_abi_emitter em_inst = load_emitter(s, r.emitter_index);
  const etx_abi_emitter_profile em = load_emitter_profile(s, em_inst.profile);
  switch (em_inst.cls) {
    case ETX_EMITTER_AREA: {
      const etx_abi_triangle& tri = s.triangles[em_inst.triangle_index];
      const etx_abi_material& material = s.materials[tri.material_index];
      r.barycentric = random_barycentric(smp.next_2d());
      Vtx vertex = lerp_vertex(s, tri, r.barycentric);
      r.origin = vertex.pos;
      r.normal = vertex.nrm;
      r.direction = sample_cosine_distribution(smp.next_2d(), r.normal, vertex.tan, vertex.btn, collimation_to_exponent(material.emission_collimation));
      // scene_emitters.hxx:21-38 emitter_evaluate_out_local
      r.pdf_dir = fmaxf(0.0f, dot(r.normal, r.direction)) * kInvPi;
      if (r.pdf_dir <= 0.0f) {
        r.value = mk3(0.0f);
      } else {
        r.pdf_area = 1.0f / em_inst.triangle_area;
        r.pdf_dir_out = r.pdf_dir * r.pdf_area;
        r.value = apply_image(s, em.emission, vertex.tex, nullptr, wavelength);
      }
      break;
    }
    case ETX_EMITTER_DIRECTIONAL: {
      f3 direction_to_scene = ld3(em.direction) * (-1.0f);
      Basis basis = orthonormal_basis(direction_to_scene);
      f2 pos_sample = sample_disk(smp.next_2d());
      f2 dir_sample = sample_disk(smp.next_2d());
      r.direction = normalize(direction_to_scene + basis.u * dir_sample.x * (0.5f * em.equivalent_disk_size) + basis.v * dir_sample.y * (0.5f * em.equivalent_disk_size));
      r.pdf_dir = 1.0f;
      r.pdf_area = env_pdf_area(s);
      r.pdf_dir_out = r.pdf_dir * r.pdf_area;
      r.normal = direction_to_scene;
      r.origin = s.bounds_center + s.bounds_radius * (pos_sample.x * basis.u + pos_sample.y * basis.v - direction_to_scene);
      r.origin += r.direction * distance_to_sphere(r.origin, r.direction, s.bounds_center, s.bounds_radius);
      r.value = apply_image(s, em.emission, dir_sample * 0.5f + f2{0.5f, 0.5f}, nullptr, wavelength);
      break;
    }
    default: {  // Environment
      const DImage& img = s.images[em.emission.image_index];
      float pdf_image = 0.0f;
      float4 image_value = make_float4(0, 0, 0, 0);
      f2 uv = image_sample(img, smp.next_2d(), pdf_image, image_value);
      if (pdf_image == 0.0f)
        return emitter_sample_zero();
      float sin_t = fmaxf(kEpsilon, sin_rev(uv.y * 0.5f));
      f3 d = -uv_to_direction(uv, img.offset, img.scale.x);
      Basis basis = orthonormal_basis(d);
      f2 disk_sample = sample_disk(smp.next_2d());
      r.direction = d;
      r.normal = d;
      r.origin = s.bounds_center + s.bounds_radius * (disk_sample.x * basis.u + disk_sample.y * basis.v - d);
      r.origin += r.direction * distance_to_sphere(r.origin, r.direction, s.bounds_center, s.bounds_radius);
      r.value = spectrum_eval(s, em.emission.spectrum_index, wavelength) * mk3(image_value);
      r.pdf_area = env_pdf_area(s);
      r.pdf_dir = pdf_image / (2.0f * kPi * kPi * sin_t);
      r.pdf_dir_out = r.pdf_area * r.pdf_dir;
      break;
    }
  }
  r.triangle_index = em_inst.triangle_index;
  r.medium_index = emitter_external_medium_index(s, em_inst);
  r.is_delta = em_inst.cls == ETX_EMITTER_DIRECTIONAL;
  r.is_distant = em_inst.cls != ETX_EMITTER_AREA;
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// camera  scene_camera.hxx

ETX_DEV f2 get_jittered_uv(Sampler& smp, uint32_t px, uint32_t py, uint32_t w, uint32_t h) {  // scene_camera.hxx:12-18
  float a = smp.next();
  float b = smp.next();
  return {(float(px) + 0.5f + 0.5f * (a * 2.0f - 1.0f)) / float(w) * 2.0f - 1.0f, (float(py) + 0.5f + 0.5f * (b * 2.0f - 1.0f)) / float(h) * 2.0f - 1.0f};
}

struct RayGen {
  f3 o, d;
  float tmin, tmax;
};

ETX_DEV RayGen generate_ray(const DScene& s, const f2 uv, const f2 sensor_rnd) {  // scene_camera.hxx:26-62
  const DCamera& c = s.camera;
  if (c.cls == 1u)
    return {c.position, from_spherical(uv.x * kPi, uv.y * kHalfPi), kRayEpsilon, kMaxFloat};
  f3 origin = c.position;
  f3 su = uv.x * c.side;
  f3 u = uv.y * c.up / c.aspect;
  f3 w_o = normalize(c.tan_half_fov * (su + u) + c.direction);
  if ((c.lens_radius > kEpsilon) && (c.focal_distance > kEpsilon)) {
    f2 sensor_sample;
    if (c.lens_image == kInvalid) {
      sensor_sample = sample_disk(sensor_rnd);
    } else {
      float pdf;
      float4 value;
      f2 t = image_sample(s.images[c.lens_image], sensor_rnd, pdf, value);
      sensor_sample = {t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f};
    }
    sensor_sample = sensor_sample * c.lens_radius;
    origin = origin + c.side * sensor_sample.x + c.up * sensor_sample.y;
    float focal_plane_distance = c.focal_distance / dot(w_o, c.direction);
    f3 p = c.position + focal_plane_distance * w_o;
    w_o = normalize(p - origin);
  }
  float cos_t = dot(w_o, c.direction);
  float t_near = c.clip_near > 0.0f ? c.clip_near / cos_t : kRayEpsilon;
  float t_far = c.clip_far > 0.0f ? c.clip_far / cos_t : kMaxFloat;
  return {origin, w_o, fmaxf(t_near, kRayEpsilon), t_far};
}

struct CameraSample {  // camera.hxx:45-58
  f3 position, normal, direction;
  f2 uv;
  float weight, pdf_dir, pdf_area, pdf_dir_out;
};

ETX_DEV CameraSample sample_film(const DScene& s, Sampler& smp, const f3& from_point) {  // scene_camera.hxx:64-118
  const DCamera& c = s.camera;
  CameraSample r;
  r.position = r.normal = r.direction = mk3(0.0f);
  r.uv = {0.0f, 0.0f};
  r.weight = r.pdf_dir = r.pdf_area = r.pdf_dir_out = 0.0f;
  if (c.cls == 1u)
    return r;
  bool thin_lens = (c.lens_radius > kEpsilon) && (c.focal_distance > kEpsilon);
  f2 sensor_sample = {0.0f, 0.0f};
  if (thin_lens) {
    if (c.lens_image == kInvalid) {
      sensor_sample = sample_disk(smp.next_2d());
    } else {
      float pdf;
      float4 value;
      f2 t = image_sample(s.images[c.lens_image], smp.next_2d(), pdf, value);
      sensor_sample = {t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f};
    }
    sensor_sample = sensor_sample * c.lens_radius;
  }
  f3 position = c.position + sensor_sample.x * c.side + sensor_sample.y * c.up;
  f3 direction = position - from_point;
  float cos_t = -dot(direction, c.direction);
  if (cos_t < 0.0f)
    return r;
  float distance_squared = dot(direction, direction);
  float distance = sqrtf(distance_squared);
  direction = direction / distance;
  cos_t /= distance;
  float focal_plane_distance = thin_lens ? c.focal_distance : 1.0f;
  f3 focus_point = position - direction * (focal_plane_distance / cos_t);
  const float* m = c.view_proj;  // column major, vector_math.hxx:33-40
  float px = m[0] * focus_point.x + m[4] * focus_point.y + m[8] * focus_point.z + m[12];
  float py = m[1] * focus_point.x + m[5] * focus_point.y + m[9] * focus_point.z + m[13];
  float pw = m[3] * focus_point.x + m[7] * focus_point.y + m[11] * focus_point.z + m[15];
  f2 uv = {px / pw, py / pw};
  if ((pw <= 0.0f) || (uv.x < -1.0f) || (uv.y < -1.0f) || (uv.x > 1.0f) || (uv.y > 1.0f))
    return r;
  float lens_area = (c.lens_radius > kEpsilon) ? kPi * sqr(c.lens_radius) : 1.0f;
  r.position = position;
  r.direction = direction;
  r.normal = c.direction;
  r.uv = uv;
  r.pdf_area = 1.0f / lens_area;
  r.pdf_dir = r.pdf_area * distance_squared / cos_t;
  r.pdf_dir_out = 1.0f / (c.area * lens_area * cos_t * cos_t * cos_t);
  float importance = r.pdf_dir_out / cos_t;
  r.weight = importance / r.pdf_dir;
  return r;
}

ETX_DEV float film_evaluate_out_pdf_dir(const DScene& s, const f3& ray_d) {  // scene_camera.hxx:120-126
  const DCamera& c = s.camera;
  float cos_t = dot(ray_d, c.direction);
  return (c.cls == 1u) ? 1.0f : 1.0f / (c.area * cos_t * cos_t * cos_t);
}

// ---------------------------------------------------------------------------------------------------------------
// media  scene_medium.hxx

ETX_DEV float phase_function(const f3& w_i, const f3& w_o, float g) {  // scene_medium.hxx:123-127
  float cos_t = dot(w_i, w_o);
  float d = 1.0f + g * g - 2.0f * g * cos_t;
  return (1.0f / (4.0f * kPi)) * (1.0f - g * g) / (d * sqrtf(d));
}

ETX_DEV f3 sample_phase_function(const f3& w_i, float g, const f2 rnd) {  // scene_medium.hxx:129-144
  float cos_theta;
  if (fabsf(g) < 1e-3f) {
    cos_theta = 1.0f - 2.0f * rnd.x;
  } else {
    float sqr_term = (1.0f - g * g) / (1.0f + g * (2.0f * rnd.x - 1.0f));
    cos_theta = (1.0f + g * g - sqr_term * sqr_term) / (2.0f * g);
  }
  float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
  Basis basis = orthonormal_basis(w_i);
  float sp, cp;
  sincos_rev(rnd.y, &sp, &cp);
  return (basis.u * cp + basis.v * sp) * sin_theta - w_i * cos_theta;
}

struct MediumSample {  // medium.hxx:25-37
  f3 weight;
  f3 pos;
  float sampled_medium_t;
  ETX_DEV bool sampled_medium() const {
    return sampled_medium_t > 0.0f;
  }
};

// scene_medium.hxx:99-121 sample_spectrum_component (RGB branch)
ETX_DEV uint32_t sample_spectrum_component(const f3& albedo, const f3& throughput, float rnd, f3& pdf) {
  f3 at = albedo * throughput;
  if ((at.x <= kEpsilon) && (at.y <= kEpsilon) && (at.z <= kEpsilon)) {
    pdf = mk3(1.0f / 3.0f);
    return uint32_t(3.0f * rnd);
  }
  pdf = at / (at.x + at.y + at.z);
  return 2u - uint32_t(rnd < pdf.x + pdf.y) - uint32_t(rnd < pdf.x);
}

// scene_medium.hxx:241-288 sample_medium, homogeneous branch (channel-selected exponential free flight)
// sample_medium heterogeneous branch, scene_medium.hxx:284-349: delta tracking against the majorant max_sigma in the
// medium's local frame. NOTE: like the reference, `pos` and `sampled_medium_t` of the result are in that LOCAL frame
// ((p - bounds.min) / extent): the reference hands result.pos to the integrators as the new ray origin unchanged.
ETX_DEV MediumSample sample_medium_heterogeneous(const DScene& s, const DMedium& m, float wavelength, Sampler& smp, const f3& pos, const f3& w_i, float max_t) {
  MediumSample r;
  r.weight = mk3(0.0f), r.pos = mk3(0.0f), r.sampled_medium_t = 0.0f;
  if (m.max_sigma <= 0.0f)
    return r;
  f3 medium_pos, medium_dir;
  float t_min = 0.0f, t_max = 0.0f;
  if (medium_intersects_bounds(m, pos, w_i, max_t, medium_pos, medium_dir, t_min, t_max) == false)
    return r;
  f3 absorption, scattering;
  medium_coefficients(s, m, wavelength, absorption, scattering);
  const f3 extinction = scattering + absorption;
  const f3 albedo = {extinction.x > 0.0f ? scattering.x / extinction.x : 0.0f, extinction.y > 0.0f ? scattering.y / extinction.y : 0.0f,
    extinction.z > 0.0f ? scattering.z / extinction.z : 0.0f};
  float t = t_min, previous_t = t_min;
  f3 accumulated = mk3(1.0f);
  while (true) {
    t -= logf(1.0f - smp.next()) / m.max_sigma;
    if (t >= t_max)
      break;
    const float distance = fmaxf(0.0f, t - previous_t);
    accumulated *= f3{expf(-extinction.x * distance), expf(-extinction.y * distance), expf(-extinction.z * distance)};
    previous_t = t;
    const float density_value = medium_sample_density(m, medium_pos + medium_dir * t);
    if (density_value * m.max_sigma == 0.0f)
      continue;
    f3 pdf;
    const uint32_t channel = sample_spectrum_component(albedo, scattering, smp.next(), pdf);
    const float sigma_t = channel == 0 ? extinction.x : (channel == 1 ? extinction.y : extinction.z);
    const float random = smp.next();
    if ((sigma_t > 0.0f) && (random < density_value)) {
      const float pdf_sum = pdf.x + pdf.y + pdf.z;
      r.weight = (pdf_sum > 0.0f) ? (scattering * accumulated) / pdf_sum : scattering * accumulated;
      r.pos = medium_pos + medium_dir * t;
      r.sampled_medium_t = t - t_min;
      return r;
    }
  }
  const float remaining = fmaxf(0.0f, t_max - previous_t);
  accumulated *= f3{expf(-extinction.x * remaining), expf(-extinction.y * remaining), expf(-extinction.z * remaining)};
  r.weight = accumulated;
  return r;
}

// `rows`: load_medium_rows(m), which the caller keeps for the phase function and the explicit-connection switch
ETX_DEV MediumSample sample_medium_homogeneous(const DScene& s, const DMedium& m, const MediumRows& rows, float wavelength, const f3& throughput, Sampler& smp, const f3& pos, const f3& w_i,
  float max_t) {
  if (rows.cls != 0u)  // Medium::Class::Heterogeneous
    return sample_medium_heterogeneous(s, m, wavelength, smp, pos, w_i, max_t);
  f3 absorption = rows.absorption, scattering = rows.scattering;
  if (s.spectral != 0u)
    medium_coefficients(s, m, wavelength, absorption, scattering);
  f3 extinction = scattering + absorption;
  f3 albedo = {extinction.x > 0.0f ? scattering.x / extinction.x : 0.0f, extinction.y > 0.0f ? scattering.y / extinction.y : 0.0f,
    extinction.z > 0.0f ? scattering.z / extinction.z : 0.0f};
  float t = 0.0f;
  f3 pdf = mk3(0.0f);
  while (t < kRayEpsilon) {
    uint32_t channel = sample_spectrum_component(albedo, throughput, smp.next(), pdf);
    float sample_t = channel == 0 ? extinction.x : (channel == 1 ? extinction.y : extinction.z);
    t = (sample_t > 0.0f) ? -logf(1.0f - smp.next()) / sample_t : max_t;
  }
  t = fminf(t, max_t);
  bool sampled = t < max_t;
  f3 tr = {expf(-t * extinction.x), expf(-t * extinction.y), expf(-t * extinction.z)};
  pdf *= sampled ? tr * extinction : tr;
  MediumSample r;
  r.pos = mk3(0.0f);
  r.sampled_medium_t = 0.0f;
  if ((pdf.x <= kEpsilon) && (pdf.y <= kEpsilon) && (pdf.z <= kEpsilon)) {
    r.weight = mk3(0.0f);
    return r;
  }
  r.pos = pos + w_i * t;
  r.sampled_medium_t = sampled ? t : 0.0f;
  r.weight = (sampled ? tr * scattering : tr) / (pdf.x + pdf.y + pdf.z);
  return r;
}
ETX_DEV MediumSample sample_medium_homogeneous(const DScene& s, const DMedium& m, float wavelength, const f3& throughput, Sampler& smp, const f3& pos, const f3& w_i, float max_t) {
  return sample_medium_homogeneous(s, m, load_medium_rows(m), wavelength, throughput, smp, pos, w_i, max_t);
}

}  // namespace etxd
