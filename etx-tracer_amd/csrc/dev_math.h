// dev_math.h - fp32 vector math, sampling helpers and the per-path RNG for the gfx950 kernels.
// Semantics follow the reference's header-only math so that estimators (not just noise) agree:
//   sources/etx/render/shared/math.hxx, sampler.hxx (cited per function).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#if !defined(ETX_DEV)
#define ETX_DEV __device__ __forceinline__
#endif
#define ETX_HD __host__ __device__ __forceinline__

namespace etxd {

constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kDoublePi = 6.283185307179586476925286766559f;
constexpr float kHalfPi = 1.5707963267948966192313216916398f;
constexpr float kQuarterPi = 0.78539816339744830961566084581988f;
constexpr float kInvPi = 0.31830988618379067153776752674503f;
constexpr float kEpsilon = 1.192092896e-07f;
constexpr float kMaxFloat = 3.402823466e+38f;
constexpr float kMaxHalf = 65504.0f;
constexpr float kInvMaxHalf = 1.0f / kMaxHalf;
constexpr float kRayEpsilon = 15.0f / (kMaxHalf - 1.0f);  // math.hxx:118
constexpr float kDeltaAlphaTreshold = 1.0e-4f;             // math.hxx:119
constexpr uint32_t kInvalid = 0xffffffffu;

struct f2 {
  float x, y;
};
struct f3 {
  float x, y, z;
};

ETX_HD f3 mk3(float x, float y, float z) {
  return f3{x, y, z};
}
ETX_HD f3 mk3(float v) {
  return f3{v, v, v};
}
ETX_HD f3 mk3(const float4& v) {
  return f3{v.x, v.y, v.z};
}
ETX_HD float4 mk4(const f3& v, float w) {
  return make_float4(v.x, v.y, v.z, w);
}
ETX_HD f3 operator+(const f3& a, const f3& b) {
  return {a.x + b.x, a.y + b.y, a.z + b.z};
}
ETX_HD f3 operator-(const f3& a, const f3& b) {
  return {a.x - b.x, a.y - b.y, a.z - b.z};
}
ETX_HD f3 operator*(const f3& a, const f3& b) {
  return {a.x * b.x, a.y * b.y, a.z * b.z};
}
ETX_HD f3 operator/(const f3& a, const f3& b) {
  return {a.x / b.x, a.y / b.y, a.z / b.z};
}
ETX_HD f3 operator*(const f3& a, float b) {
  return {a.x * b, a.y * b, a.z * b};
}
ETX_HD f3 operator*(float b, const f3& a) {
  return {a.x * b, a.y * b, a.z * b};
}
ETX_HD f3 operator/(const f3& a, float b) {
  return {a.x / b, a.y / b, a.z / b};
}
ETX_HD f3 operator-(const f3& a) {
  return {-a.x, -a.y, -a.z};
}
ETX_HD f3& operator+=(f3& a, const f3& b) {
  a.x += b.x, a.y += b.y, a.z += b.z;
  return a;
}
ETX_HD f3& operator*=(f3& a, const f3& b) {
  a.x *= b.x, a.y *= b.y, a.z *= b.z;
  return a;
}
ETX_HD f3& operator*=(f3& a, float b) {
  a.x *= b, a.y *= b, a.z *= b;
  return a;
}
ETX_HD f2 operator+(const f2& a, const f2& b) {
  return {a.x + b.x, a.y + b.y};
}
ETX_HD f2 operator*(const f2& a, float b) {
  return {a.x * b, a.y * b};
}
ETX_HD float dot(const f3& a, const f3& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
ETX_HD f3 cross(const f3& a, const f3& b) {  // math.hxx:537-543
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
ETX_HD float length(const f3& v) {
  return sqrtf(dot(v, v));
}
ETX_HD f3 normalize(const f3& v) {  // math.hxx:529-531 (true division, as the reference)
  return v / length(v);
}
ETX_HD f3 reflect(const f3& v, const f3& n) {  // math.hxx:533-535
  return v - (2.0f * dot(v, n)) * n;
}
ETX_HD f3 fmin3(const f3& a, const f3& b) {
  return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)};
}
ETX_HD f3 fmax3(const f3& a, const f3& b) {
  return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)};
}
ETX_HD float max_component(const f3& a) {
  return fmaxf(a.x, fmaxf(a.y, a.z));
}
ETX_HD float sqr(float a) {
  return a * a;
}
ETX_HD float saturate(float v) {
  return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}
ETX_HD float lerpf(float a, float b, float t) {  // math.hxx:665-668 (a*(1-t) + b*t)
  return a * (1.0f - t) + b * t;
}
ETX_HD float luminance(const f3& v) {  // math.hxx:728-730
  return v.x * 0.212671f + v.y * 0.715160f + v.z * 0.072169f;
}
ETX_HD bool valid_value(float t) {  // math.hxx:835-837
  return (t >= 0.0f) && isfinite(t);
}

// math.hxx:736-746
struct Basis {
  f3 u, v;
};
ETX_HD Basis orthonormal_basis(const f3& n) {
  f3 a = normalize(((n.x != n.y) || (n.x != n.z)) ? f3{n.z - n.y, n.x - n.z, +n.y - n.x} : f3{n.z - n.y, n.x + n.z, -n.y - n.x});
  f3 b = normalize(cross(n, a));
  return {a, b};
}

// sin / cos of an angle given in REVOLUTIONS (angle / 2 pi). v_sin_f32 / v_cos_f32 take exactly this argument (valid for
// |rev| <= 256, ~1e-6 absolute error); sincosf() costs ~150 VALU instructions in its range reduction, these cost two.
// Sampling directions and emitter pdfs only: the bit-exact paths (Sampler, offset_ray, cell indices) do not use them.
ETX_HD void sincos_rev(float rev, float* s, float* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  *s = __builtin_amdgcn_sinf(rev);
  *c = __builtin_amdgcn_cosf(rev);
#else
  sincosf(rev * kDoublePi, s, c);
#endif
}
ETX_HD float sin_rev(float rev) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sinf(rev);
#else
  return sinf(rev * kDoublePi);
#endif
}

// math.hxx:748-762 (exponent form; exponent 1 = cosine weighted)
ETX_HD f3 sample_cosine_distribution(const f2 rnd, float exponent) {
  const float x = fmaxf(rnd.x, kEpsilon);
  float cos_theta = (exponent == 1.0f) ? sqrtf(x) : powf(x, 1.0f / (exponent + 1.0f));  // the cosine lobe (every call site but one) folds to the sqrt
  float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
  float s, c;
  sincos_rev(rnd.y, &s, &c);
  return {c * sin_theta, s * sin_theta, cos_theta};
}
ETX_HD f3 sample_cosine_distribution(const f2 rnd, const f3& n, const f3& u, const f3& v, float exponent) {
  f3 l = sample_cosine_distribution(rnd, exponent);
  return u * l.x + v * l.y + n * l.z;
}
ETX_HD f3 sample_cosine_distribution(const f2 rnd, const f3& n, float exponent) {
  Basis b = orthonormal_basis(n);
  return sample_cosine_distribution(rnd, n, b.u, b.v, exponent);
}

ETX_HD f3 barycentrics(float u, float v) {  // math.hxx:764-766
  return {1.0f - u - v, u, v};
}
ETX_HD f3 random_barycentric(const f2 rnd) {  // math.hxx:768-771
  float r1 = sqrtf(rnd.x);
  return {1.0f - r1, r1 * (1.0f - rnd.y), r1 * rnd.y};
}

// math.hxx:773-790 (concentric disk)
ETX_HD f2 sample_disk(const f2 rnd) {
  f2 offset = {2.0f * rnd.x - 1.0f, 2.0f * rnd.y - 1.0f};
  if ((offset.x == 0.0f) && (offset.y == 0.0f))
    return {0.0f, 0.0f};
  float r, theta;
  if (fabsf(offset.x) > fabsf(offset.y)) {
    r = offset.x;
    theta = kQuarterPi * (offset.y / offset.x);
  } else {
    r = offset.y;
    theta = kHalfPi - kQuarterPi * (offset.x / offset.y);
  }
  float s, c;
  sincos_rev(theta * (1.0f / kDoublePi), &s, &c);
  return {r * c, r * s};
}

// math.hxx:810-822 (projecected_coords / disk_uv)
ETX_HD f2 disk_uv(const f3& normal, const f3& in_dir, float sz, float csz) {
  f2 pc = {0.0f, 0.0f};
  if (sz != 0.0f) {
    Basis b = orthonormal_basis(normal);
    pc = {dot(b.u, in_dir) / (0.5f * sz * csz), dot(b.v, in_dir) / (0.5f * sz * csz)};
  }
  return {saturate(pc.x * 0.5f + 0.5f), saturate(pc.y * 0.5f + 0.5f)};
}

// math.hxx:925-943 : origin offset with the integer-ULP trick
ETX_HD f3 offset_ray(const f3& p, const f3& n) {
  constexpr float int_scale = 256.0f;
  constexpr float float_scale = 1.0f / 65536.0f;
  constexpr float origin = 1.0f / 32.0f;
  int32_t of_i_x = static_cast<int32_t>(int_scale * n.x);
  int32_t of_i_y = static_cast<int32_t>(int_scale * n.y);
  int32_t of_i_z = static_cast<int32_t>(int_scale * n.z);
  float p_i_x = __int_as_float(__float_as_int(p.x) + ((p.x > 0.0f) ? of_i_x : -of_i_x));
  float p_i_y = __int_as_float(__float_as_int(p.y) + ((p.y > 0.0f) ? of_i_y : -of_i_y));
  float p_i_z = __int_as_float(__float_as_int(p.z) + ((p.z > 0.0f) ? of_i_z : -of_i_z));
  return {
    fabsf(p.x) < origin ? p.x + float_scale * n.x : p_i_x,
    fabsf(p.y) < origin ? p.y + float_scale * n.y : p_i_y,
    fabsf(p.z) < origin ? p.z + float_scale * n.z : p_i_z,
  };
}

ETX_HD float power_heuristic(float f, float g) {  // math.hxx:945-950
  float f2v = f * f, g2 = g * g;
  float denom = f2v + g2;
  return denom > 0.0f ? saturate(f2v / denom) : 0.0f;
}

// math.hxx:952-974
ETX_HD f3 from_spherical(float phi, float theta) {
  float sp, cp, st, ct;
  sincos_rev(phi * (1.0f / kDoublePi), &sp, &cp);
  sincos_rev(theta * (1.0f / kDoublePi), &st, &ct);
  return {cp * ct, st, sp * ct};
}
// math.hxx:976-998
ETX_HD f3 uv_to_direction(const f2 uv, const f2 offset, float u_scale) {
  float u = uv.x;
  if (u_scale < 0.0f)
    u = 1.0f - u;
  u = u - offset.x;
  u = u - floorf(u);
  float phi = (u * 2.0f - 1.0f) * kPi;
  float theta = (0.5f - uv.y) * kPi;
  return from_spherical(phi, theta);
}
ETX_HD f2 direction_to_uv(const f3& dir, const f2 offset, float u_scale) {
  float r = length(dir);
  float phi = atan2f(dir.z, dir.x);
  float theta = asinf(dir.y / r);
  float u = (phi / kPi + 1.0f) / 2.0f;
  if (u_scale < 0.0f)
    u = 1.0f - u;
  u = u + offset.x;
  u = u - floorf(u);
  return {u, 0.5f - theta / kPi};
}

// math.hxx:1024-1037
ETX_HD float distance_to_sphere(const f3& r_origin, const f3& r_direction, const f3& center, float radius) {
  f3 e = r_origin - center;
  float b = dot(r_direction, e);
  float d = (b * b) - dot(e, e) + (radius * radius);
  if (d < 0.0f)
    return 0.0f;
  d = sqrtf(d);
  float a0 = -b - d;
  float a1 = -b + d;
  return (a0 < 0.0f) ? ((a1 < 0.0f) ? 0.0f : a1) : a0;
}

// ---------------------------------------------------------------------------------------------------------------
// Sampler  sources/etx/render/shared/sampler.hxx:7-78
// init(a,b) = 16-round TEA of (path index, iteration); next() = integer hash -> mantissa trick in [0,1).
// fixed_u/v/w carry three pre-drawn numbers into BSDF sampling (push_fixed / pop_fixed / has_fixed).
struct Sampler {
  uint32_t seed;
  float fixed_u, fixed_v, fixed_w;

  ETX_HD static uint32_t random_seed(uint32_t val0, uint32_t val1) {
    uint32_t v0 = val0, v1 = val1, s0 = 0u;
#pragma unroll
    for (uint32_t n = 0u; n < 16u; ++n) {
      s0 += 0x9e3779b9u;
      v0 += ((v1 << 4u) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5u) + 0xc8013ea4u);
      v1 += ((v0 << 4u) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5u) + 0x7e95761eu);
    }
    return v0;
  }
  ETX_HD void init(uint32_t a, uint32_t b) {
    seed = random_seed(a, b);
    fixed_u = fixed_v = fixed_w = 0.0f;
  }
  ETX_HD float next() {
    seed = (seed ^ 61u) ^ (seed >> 16u);
    seed *= 9u;
    seed = seed ^ (seed >> 4u);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15u);
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float((seed >> 9) | 0x3f800000u) - 1.0f;
#else
    union {
      uint32_t i;
      float f;
    } w = {(seed >> 9) | 0x3f800000u};
    return w.f - 1.0f;
#endif
  }
  ETX_HD f2 next_2d() {
    float a = next();
    float b = next();
    return {a, b};
  }
  ETX_HD void push_fixed(float u, float v, float w) {
    fixed_u = u, fixed_v = v, fixed_w = w;
  }
  ETX_HD void pop_fixed() {
    fixed_u = fixed_v = fixed_w = 0.0f;
  }
  ETX_HD bool has_fixed() const {
    return (sqr(fixed_u) + sqr(fixed_v) + sqr(fixed_w)) > kEpsilon;
  }
};

}  // namespace etxd
