/* ORACLE / TEST INFRASTRUCTURE ONLY - never linked into libetx_hip.so.
 *
 * Plain-C restatement of the small deterministic building blocks of the hot path, used as the checker for the
 * device-side known-answer kernels (etx_hip_kat). Each function cites the reference lines it follows; the
 * restatement is pinned against values produced by the reference's own headers (oracle/_ref/etx_oracle --kat ->
 * tests/golden/kat_reference.json, see oracle/gen_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* sources/etx/render/shared/sampler.hxx:54-64  Sampler::random_seed (16 round TEA) */
uint32_t kat_random_seed(uint32_t val0, uint32_t val1) {
  uint32_t v0 = val0, v1 = val1, s0 = 0u;
  for (uint32_t n = 0u; n < 16u; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4u) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5u) + 0xc8013ea4u);
    v1 += ((v0 << 4u) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5u) + 0x7e95761eu);
  }
  return v0;
}

/* sampler.hxx:66-77  Sampler::next_random */
float kat_next_random(uint32_t* seed) {
  uint32_t s = *seed;
  s = (s ^ 61u) ^ (s >> 16u);
  s *= 9u;
  s = s ^ (s >> 4u);
  s *= 0x27d4eb2du;
  s = s ^ (s >> 15u);
  *seed = s;
  uint32_t bits = (s >> 9) | 0x3f800000u;
  float f;
  memcpy(&f, &bits, 4);
  return f - 1.0f;
}

/* out: seed, next, next, next */
void kat_sampler(uint32_t a, uint32_t b, float out[4]) {
  uint32_t seed = kat_random_seed(a, b);
  memcpy(&out[0], &seed, 4);
  out[1] = kat_next_random(&seed);
  out[2] = kat_next_random(&seed);
  out[3] = kat_next_random(&seed);
}

/* sources/etx/render/shared/math.hxx:925-943  offset_ray */
void kat_offset_ray(const float p[3], const float n[3], float out[3]) {
  const float int_scale = 256.0f, float_scale = 1.0f / 65536.0f, origin = 1.0f / 32.0f;
  for (int i = 0; i < 3; ++i) {
    int32_t of_i = (int32_t)(int_scale * n[i]);
    int32_t pi;
    memcpy(&pi, &p[i], 4);
    pi += (p[i] > 0.0f) ? of_i : -of_i;
    float p_i;
    memcpy(&p_i, &pi, 4);
    out[i] = fabsf(p[i]) < origin ? p[i] + float_scale * n[i] : p_i;
  }
}

static void normalize3(float v[3]) {
  float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= l, v[1] /= l, v[2] /= l;
}

/* math.hxx:736-746  orthonormal_basis ; out: u.xyz, v.xyz */
void kat_orthonormal_basis(const float n[3], float out[6]) {
  float a[3];
  if ((n[0] != n[1]) || (n[0] != n[2])) {
    a[0] = n[2] - n[1], a[1] = n[0] - n[2], a[2] = n[1] - n[0];
  } else {
    a[0] = n[2] - n[1], a[1] = n[0] + n[2], a[2] = -n[1] - n[0];
  }
  normalize3(a);
  float b[3] = {n[1] * a[2] - a[1] * n[2], n[2] * a[0] - a[2] * n[0], n[0] * a[1] - a[0] * n[1]};
  normalize3(b);
  memcpy(out, a, 12);
  memcpy(out + 3, b, 12);
}

/* math.hxx:748-762  sample_cosine_distribution(rnd, n, exponent = 1) */
void kat_sample_cosine(const float rnd[2], const float n[3], float out[3]) {
  const float kEpsilon = 1.192092896e-07f, kDoublePi = 6.283185307179586476925286766559f;
  float cos_theta = powf(fmaxf(rnd[0], kEpsilon), 1.0f / 2.0f);
  float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
  float l[3] = {cosf(rnd[1] * kDoublePi) * sin_theta, sinf(rnd[1] * kDoublePi) * sin_theta, cos_theta};
  float uv[6];
  kat_orthonormal_basis(n, uv);
  for (int i = 0; i < 3; ++i)
    out[i] = uv[i] * l[0] + uv[3 + i] * l[1] + n[i] * l[2];
}

/* sources/etx/rt/shared/vcm_shared.hxx:820-822  VCMSpatialGridData::cell_index */
uint32_t kat_cell_index(int32_t x, int32_t y, int32_t z, uint32_t mask) {
  return (((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349663u) ^ ((uint32_t)z * 83492791u)) & mask;
}

/* math.hxx:773-790  sample_disk */
void kat_sample_disk(const float rnd[2], float out[2]) {
  const float kQuarterPi = 0.78539816339744830961566084581988f, kHalfPi = 1.5707963267948966192313216916398f;
  float ox = 2.0f * rnd[0] - 1.0f, oy = 2.0f * rnd[1] - 1.0f;
  if ((ox == 0.0f) && (oy == 0.0f)) {
    out[0] = out[1] = 0.0f;
    return;
  }
  float r, theta;
  if (fabsf(ox) > fabsf(oy)) {
    r = ox;
    theta = kQuarterPi * (oy / ox);
  } else {
    r = oy;
    theta = kHalfPi - kQuarterPi * (ox / oy);
  }
  out[0] = r * cosf(theta);
  out[1] = r * sinf(theta);
}

/* vcm_cpu.cxx:100-113  radius schedule and VC/VM weights of iteration `it`; out: radius, vc_weight, vm_weight, vm_normalization */
void kat_vcm_iteration(float initial_radius, float scene_radius, uint32_t max_dim, uint32_t radius_decay, uint32_t it, uint32_t pixel_count, int merging, float out[4]) {
  const float kPi = 3.1415926535897932384626433832795f;
  float used_radius = initial_radius;
  if (used_radius == 0.0f)
    used_radius = 5.0f * scene_radius / (float)max_dim;
  float radius_scale = 1.0f / (1.0f + (float)it / (float)radius_decay);
  float r = used_radius * radius_scale;
  float eta_vcm = kPi * (r * r) * (float)pixel_count;
  out[0] = r;
  out[1] = 1.0f / eta_vcm;
  out[2] = merging ? eta_vcm : 0.0f;
  out[3] = 1.0f / eta_vcm;
}
