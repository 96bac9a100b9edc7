// dev_bsdf_ool.h - out-of-line entry points of the general scattering models.
//
// The general-material kernels used to inline the whole BSDF library (all eleven classes of scene_bsdf.hxx:56-107, the
// Heitz multiple-scattering walks of bsdf_external.hxx, thin-film Fresnel) at every call site of sample / evaluate / pdf /
// reverse_pdf: five call sites per step function, two copies of the class switch per site (PrincipledBSDF evaluates one of
// three classes on a modified material copy) - 1 M lines of ISA per kernel, 512 VGPRs, hundreds of spills, 13-33 min of
// compile time per translation unit. Here every heavy class function is ONE real function per translation unit
// (noinline): the kernels call it with everything BY VALUE - no pointer to a caller's private memory crosses a call, the
// scene is reached through its device-resident copy (DScene::self), the material through its index, the sampler
// returns its advanced seed in the result. The register budget of a kernel is then max(caller, callee) instead of the
// sum over every inlined copy, and the callee's budget is bounded by one class.
//
// PrincipledBSDF (bsdf_principled.hxx:16-114) picks Conductor / Dielectric / Plastic per call and evaluates it on a
// modified copy of the material; the three modified copies are static, so the host appends them to the material table
// at upload (DScene::material_variants) and the pick becomes an index - no 200-byte material copy in private memory.
#pragma once

#include "dev_bsdf.h"

namespace etxd {

#define ETX_OOL static __device__ __attribute__((noinline))

struct SampleRet {
  BsdfSample s;
  uint32_t seed;
};
struct EvalRet {
  BsdfEval e;
  uint32_t seed;
};
struct PdfRet {
  float pdf;
  uint32_t seed;
};

// ---- sample
ETX_OOL SampleRet ool_diffuse_sample(const DScene* sp, BsdfData d, uint32_t mi, Sampler smp) {
  const BsdfSample r = diffuse_sample(*sp, d, sp->materials[mi], smp);
  return {r, smp.seed};
}
ETX_OOL SampleRet ool_conductor_sample(const DScene* sp, BsdfData d, uint32_t mi, Sampler smp) {
  const BsdfSample r = conductor_sample(*sp, d, sp->materials[mi], smp);
  return {r, smp.seed};
}
ETX_OOL SampleRet ool_dielectric_sample(const DScene* sp, BsdfData d, uint32_t mi, Sampler smp) {
  const BsdfSample r = dielectric_sample(*sp, d, sp->materials[mi], smp);
  return {r, smp.seed};
}
ETX_OOL SampleRet ool_thinfilm_sample(const DScene* sp, BsdfData d, uint32_t mi, Sampler smp) {
  const BsdfSample r = thinfilm_sample(*sp, d, sp->materials[mi], smp);
  return {r, smp.seed};
}
ETX_OOL SampleRet ool_plastic_sample(const DScene* sp, BsdfData d, uint32_t mi, Sampler smp) {
  const BsdfSample r = plastic_sample(*sp, d, sp->materials[mi], smp);
  return {r, smp.seed};
}
ETX_OOL SampleRet ool_velvet_sample(const DScene* sp, BsdfData d, uint32_t mi, Sampler smp) {
  const BsdfSample r = velvet_sample(*sp, d, sp->materials[mi], smp);
  return {r, smp.seed};
}

// ---- evaluate
ETX_OOL EvalRet ool_diffuse_evaluate(const DScene* sp, BsdfData d, f3 w_o, uint32_t mi, Sampler smp) {
  const BsdfEval e = diffuse_evaluate_v(*sp, d, w_o, sp->materials[mi], smp);
  return {e, smp.seed};
}
ETX_OOL EvalRet ool_conductor_evaluate(const DScene* sp, BsdfData d, f3 w_o, uint32_t mi, Sampler smp) {
  const BsdfEval e = conductor_evaluate(*sp, d, w_o, sp->materials[mi], smp);
  return {e, smp.seed};
}
ETX_OOL EvalRet ool_dielectric_evaluate(const DScene* sp, BsdfData d, f3 w_o, uint32_t mi, Sampler smp) {
  const BsdfEval e = dielectric_evaluate(*sp, d, w_o, sp->materials[mi], smp);
  return {e, smp.seed};
}
ETX_OOL EvalRet ool_plastic_evaluate(const DScene* sp, BsdfData d, f3 w_o, uint32_t mi, Sampler smp) {
  const BsdfEval e = plastic_evaluate(*sp, d, w_o, sp->materials[mi], smp);
  return {e, smp.seed};
}
ETX_OOL EvalRet ool_velvet_evaluate(const DScene* sp, BsdfData d, f3 w_o, uint32_t mi) {
  const BsdfEval e = velvet_evaluate(*sp, d, w_o, sp->materials[mi]);
  return {e, 0u};
}

// ---- pdf
ETX_OOL PdfRet ool_dielectric_pdf(const DScene* sp, BsdfData d, f3 w_o, uint32_t mi, Sampler smp) {
  const float p = dielectric_pdf(*sp, d, w_o, sp->materials[mi], smp);
  return {p, smp.seed};
}
ETX_OOL PdfRet ool_plastic_pdf(const DScene* sp, BsdfData d, f3 w_o, uint32_t mi, Sampler smp) {
  const float p = plastic_pdf(*sp, d, w_o, sp->materials[mi], smp);
  return {p, smp.seed};
}

// PrincipledBSDF::sample / evaluate / pdf (bsdf_principled.hxx:24-114): metalness and transmission pick the class with
// the path's sampler, once per call; the modified material is variant 0 / 1 / 2 behind DScene::material_variants.
ETX_DEV uint32_t resolve_material(const DScene& s, const BsdfData& d, uint32_t mi, Sampler& smp) {
  const etx_abi_material& m = s.materials[mi];
  if (m.cls != ETX_MAT_PRINCIPLED)
    return mi;
  const uint32_t base = s.material_variants[mi];
  const float metalness = m.metalness.value.x * evaluate_image(s, m.metalness, d.tex, 1.0f);  // evaluate_metalness, scene.hxx:283-285
  if (smp.next() < metalness)
    return base + kPrincipledConductor;
  if (smp.next() < m.transmission.value.x)
    return base + kPrincipledDielectric;
  return base + kPrincipledPlastic;
}

// scene_bsdf.hxx:56-68 bsdf::sample over all classes
ETX_DEV BsdfSample bsdf_sample_general(const DScene& s, const BsdfData& d, uint32_t in_mi, Sampler& smp) {
  const uint32_t mi = resolve_material(s, d, in_mi, smp);
  const etx_abi_material& m = s.materials[mi];
  SampleRet r;
  switch (m.cls) {
    case ETX_MAT_DIFFUSE:
      if (m.diffuse_variation == 0u)
        return diffuse_sample_lambert(s, d, m, smp);
      r = ool_diffuse_sample(s.self, d, mi, smp);
      break;
    case ETX_MAT_TRANSLUCENT:
      return translucent_sample(s, d, m, smp);
    case ETX_MAT_CONDUCTOR:
      r = ool_conductor_sample(s.self, d, mi, smp);
      break;
    case ETX_MAT_DIELECTRIC:
      r = ool_dielectric_sample(s.self, d, mi, smp);
      break;
    case ETX_MAT_THINFILM:
      r = ool_thinfilm_sample(s.self, d, mi, smp);
      break;
    case ETX_MAT_PLASTIC:
      r = ool_plastic_sample(s.self, d, mi, smp);
      break;
    case ETX_MAT_VELVET:
      r = ool_velvet_sample(s.self, d, mi, smp);
      break;
    case ETX_MAT_MIRROR:
    case ETX_MAT_BOUNDARY:
      return bsdf_sample_delta_classes(s, d, m);
    default: {  // Void, bsdf_various.hxx:5-15
      BsdfSample v = sample_zero();
      v.w_o = d.w_i;
      v.properties = kSampleDelta;
      v.medium_index = d.medium;
      return v;
    }
  }
  smp.seed = r.seed;
  return r.s;
}

// scene_bsdf.hxx:70-80 bsdf::evaluate
ETX_DEV BsdfEval bsdf_evaluate_general(const DScene& s, const BsdfData& d, const f3& w_o, uint32_t in_mi, Sampler& smp) {
  const uint32_t mi = resolve_material(s, d, in_mi, smp);
  const etx_abi_material& m = s.materials[mi];
  EvalRet r;
  switch (m.cls) {
    case ETX_MAT_DIFFUSE:
      if (m.diffuse_variation == 0u)
        return diffuse_evaluate(s, d, w_o, m);
      r = ool_diffuse_evaluate(s.self, d, w_o, mi, smp);
      break;
    case ETX_MAT_TRANSLUCENT:
      return translucent_evaluate(s, d, w_o, m);
    case ETX_MAT_CONDUCTOR:
      r = ool_conductor_evaluate(s.self, d, w_o, mi, smp);
      break;
    case ETX_MAT_DIELECTRIC:
      r = ool_dielectric_evaluate(s.self, d, w_o, mi, smp);
      break;
    case ETX_MAT_PLASTIC:
      r = ool_plastic_evaluate(s.self, d, w_o, mi, smp);
      break;
    case ETX_MAT_VELVET:
      return ool_velvet_evaluate(s.self, d, w_o, mi).e;
    case ETX_MAT_MIRROR: {  // bsdf_various.hxx:226-240
      BsdfEval e = eval_zero();
      Frame frame = normal_frame(d);
      if (direction_matches(normalize(reflect(d.w_i, frame.nrm)), normalize(w_o))) {
        e.func = apply_image(s, m.scattering, d.tex, nullptr, d.wavelength);
        e.bsdf = e.func;
        e.pdf = 1.0f;
      }
      return e;
    }
    default:  // Boundary, Void, Thinfilm: bsdf_various.hxx:272-274, 17-19, bsdf_dielectric.hxx:43-45
      return eval_zero();
  }
  smp.seed = r.seed;
  return r.e;
}

// scene_bsdf.hxx:94-104 bsdf::pdf
ETX_DEV float bsdf_pdf_general(const DScene& s, const BsdfData& d, const f3& w_o, uint32_t in_mi, Sampler& smp) {
  const uint32_t mi = resolve_material(s, d, in_mi, smp);
  const etx_abi_material& m = s.materials[mi];
  PdfRet r;
  switch (m.cls) {
    case ETX_MAT_DIFFUSE:
      return diffuse_pdf(d, w_o);
    case ETX_MAT_TRANSLUCENT:
      return translucent_pdf(s, d, w_o, m);
    case ETX_MAT_CONDUCTOR:
      return conductor_pdf(s, d, w_o, m);
    case ETX_MAT_DIELECTRIC:
      r = ool_dielectric_pdf(s.self, d, w_o, mi, smp);
      break;
    case ETX_MAT_PLASTIC:
      r = ool_plastic_pdf(s.self, d, w_o, mi, smp);
      break;
    case ETX_MAT_VELVET:
      return velvet_pdf(d);
    case ETX_MAT_MIRROR: {
      Frame frame = normal_frame(d);
      return direction_matches(normalize(reflect(d.w_i, frame.nrm)), normalize(w_o)) ? 1.0f : 0.0f;
    }
    default:
      return 0.0f;
  }
  smp.seed = r.seed;
  return r.pdf;
}

// ---------------------------------------------------------------------------------------------------------------
// What the step functions call: <true> = the classes of the simple-material kernels, inline (dev_bsdf.h);
// <false> = every class, heavy ones out of line.
template <bool kSimple>
ETX_DEV BsdfSample bsdf_sample_s(const DScene& s, const BsdfData& d, const etx_abi_material& m, Sampler& smp) {
  if (kSimple)
    return bsdf_sample_simple(s, d, m, smp);
  return bsdf_sample_general(s, d, uint32_t(&m - s.materials), smp);
}
template <bool kSimple>
ETX_DEV BsdfEval bsdf_evaluate_s(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m, Sampler& smp) {
  if (kSimple)
    return bsdf_evaluate_simple(s, d, w_o, m);
  return bsdf_evaluate_general(s, d, w_o, uint32_t(&m - s.materials), smp);
}
template <bool kSimple>
ETX_DEV float bsdf_pdf_s(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m, Sampler& smp) {
  if (kSimple)
    return bsdf_pdf_simple(s, d, w_o, m);
  return bsdf_pdf_general(s, d, w_o, uint32_t(&m - s.materials), smp);
}
// scene_bsdf.hxx:82-92 reverse_pdf: swap the roles of w_i and w_o
template <bool kSimple>
ETX_DEV float bsdf_reverse_pdf_s(const DScene& s, const BsdfData& in_d, const f3& in_w_o, const etx_abi_material& m, Sampler& smp) {
  BsdfData d = in_d;
  f3 w_o = -in_d.w_i;
  d.w_i = -in_w_o;
  return bsdf_pdf_s<kSimple>(s, d, w_o, m, smp);
}

}  // namespace etxd
