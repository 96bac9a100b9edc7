// dev_vcm.h - per-path VCM logic (RGB mode), restated from sources/etx/rt/shared/vcm_shared.hxx for the wavefront
// kernels in kernels_vcm.hip. One lane = one path; every function cites the reference lines it follows.
#pragma once

#include "dev_bvh.h"
#include "dev_bsdf_ool.h"
#include "dev_emitters.h"
#include "dev_sss.h"
#include "pipeline.h"

namespace etxd {

// ---------------------------------------------------------------------------------------------------------------
// Cell-ordered photon merge (kernels_connect.hip): camera vertices are counting-sorted by a coarse spatial bucket. The predicate and
// the bucket are shared by the histogram (store_camera_vertex, dev_vcm_steps.h) and the scatter (k_merge_scatter).
ETX_DEV uint32_t spread_bits_6(uint32_t v) {  // 6 bits -> every third bit
  v &= 0x3fu;
  v = (v | (v << 8u)) & 0x300fu;
  v = (v | (v << 4u)) & 0x30c3u;
  v = (v | (v << 2u)) & 0x9249u;
  return v;
}

ETX_DEV uint32_t merge_bucket(const GridParams& g, const f3& pos) {
  // coarse block coordinates: the scene extent maps to 64 blocks per axis
  f3 ext = g.bbox_max - g.bbox_min;
  float scale = float(1u << kMergeBucketBits) / fmaxf(fmaxf(ext.x, ext.y), fmaxf(ext.z, g.cell_size));
  f3 q = (pos - g.bbox_min) * scale;
  uint32_t x = min(uint32_t(fmaxf(q.x, 0.0f)), (1u << kMergeBucketBits) - 1u);
  uint32_t y = min(uint32_t(fmaxf(q.y, 0.0f)), (1u << kMergeBucketBits) - 1u);
  uint32_t z = min(uint32_t(fmaxf(q.z, 0.0f)), (1u << kMergeBucketBits) - 1u);
  return spread_bits_6(x) | (spread_bits_6(y) << 1u) | (spread_bits_6(z) << 2u);
}

// VCMSpatialGridData::gather's early-outs (vcm_shared.hxx:886-893) on a stored camera vertex: pos_info.w = (depth << 8) | kCv* flags
ETX_DEV bool merge_candidate(const GridParams& g, uint32_t max_path_length, uint32_t info, const f3& pos) {
  if ((g.valid == 0u) || (g.photon_count == 0u) || (info & (kCvMedium | kCvNoMerge)) || ((info >> 8u) + 1u > max_path_length))
    return false;
  return (pos.x >= g.bbox_min.x) && (pos.y >= g.bbox_min.y) && (pos.z >= g.bbox_min.z) && (pos.x <= g.bbox_max.x) && (pos.y <= g.bbox_max.y) && (pos.z <= g.bbox_max.z);
}

struct PathState {  // VCMPathState (vcm_shared.hxx:91-150) minus the per-pixel accumulators (they live in the film)
  f3 ray_o;
  float ray_tmin;
  f3 ray_d;
  float ray_tmax;
  f3 throughput;
  float eta;
  float d_vcm, d_vc, d_vm, path_distance;
  Sampler sampler;
  uint32_t depth;   // total_path_depth
  uint32_t medium;  // medium_index
  uint32_t flags;
  uint32_t id;      // global_index
  float wavelength; // spect.wavelength (spectral mode)
};

ETX_DEV PathState load_path(const PathSet& set, uint32_t i) {
  PathState s;
  float4 a = set.ray_o_tmin[i], b = set.ray_d_tmax[i], c = set.thr_eta[i], d = set.mis[i];
  uint4 m = set.meta[i];
  s.ray_o = {a.x, a.y, a.z}, s.ray_tmin = a.w;
  s.ray_d = {b.x, b.y, b.z}, s.ray_tmax = b.w;
  s.throughput = {c.x, c.y, c.z}, s.eta = c.w;
  s.d_vcm = d.x, s.d_vc = d.y, s.d_vm = d.z, s.path_distance = d.w;
  s.sampler.seed = m.x, s.sampler.fixed_u = s.sampler.fixed_v = s.sampler.fixed_w = 0.0f;
  s.depth = m.y, s.medium = m.z, s.flags = m.w;
  s.id = set.path_id[i];
  s.wavelength = set.wavelength[i];
  return s;
}

ETX_DEV void store_path(const PathSet& set, uint32_t i, const PathState& s) {
  set.ray_o_tmin[i] = mk4(s.ray_o, s.ray_tmin);
  set.ray_d_tmax[i] = mk4(s.ray_d, s.ray_tmax);
  set.thr_eta[i] = mk4(s.throughput, s.eta);
  set.mis[i] = make_float4(s.d_vcm, s.d_vc, s.d_vm, s.path_distance);
  set.meta[i] = make_uint4(s.sampler.seed, s.depth, s.medium, s.flags);
  set.path_id[i] = s.id;
  set.wavelength[i] = s.wavelength;
}

// Dense slot for every lane with `alive` set; one atomic per wavefront (ballot + popcount prefix).
// Must be called from wave-uniform control flow.
ETX_DEV uint32_t wave_compact_slot(bool alive, uint32_t* counter) {
  const uint64_t mask = __ballot(alive);
  const uint32_t lane = __lane_id();
  const uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
  uint32_t base = 0;
  if (lane == 0)
    base = mask ? atomicAdd(counter, uint32_t(__popcll(mask))) : 0u;
  base = __shfl(base, 0);
  return base + prefix;
}

// The same for a whole workgroup: ONE atomic per 256 threads. All atomics on a counter are serialised by its L2
// channel at ~11 ns each (pipeline.h), and a 1080p iteration used to issue ~1 M of them per counter from the shade
// kernels alone - more time than the shading itself. Must be called from workgroup-uniform control flow
// (ETX_BLOCK_LOOP); three barriers, the last one makes the scratch reusable by the next call.
struct BlockScratch {  // in LDS
  uint32_t wave_total[kBlockSize / 64u];
  uint32_t base;
};

ETX_DEV uint32_t block_compact_slot(bool alive, uint32_t* counter, BlockScratch& scratch) {
  const uint64_t mask = __ballot(alive);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6u;
  const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32u), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
  if (lane == 0u)
    scratch.wave_total[wave] = uint32_t(__popcll(mask));
  __syncthreads();
  if (threadIdx.x == 0u) {
    uint32_t total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
      total += scratch.wave_total[w];
    scratch.base = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  uint32_t slot = scratch.base + prefix;
#pragma unroll
  for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
    slot += (w < wave) ? scratch.wave_total[w] : 0u;
  __syncthreads();
  return slot;
}

// Three reservations at once: the step functions of the simple shading group need a vertex slot, a shadow-queue slot and a slot in the next path
// set per lane. Taken one after the other (above) a step met nine barriers and waited three times for an atomic's round trip to the L2; here the
// three counters are scanned together - the same three barriers ONCE, thread 0 issues the three atomics back to back.
struct BlockScratch3 {  // in LDS
  uint32_t wave_total[3][kBlockSize / 64u];
  uint32_t base[3];
};
struct Slots3 {
  uint32_t a, b, c;
};

ETX_DEV Slots3 block_compact_slot3(bool fa, uint32_t* ca, bool fb, uint32_t* cb, bool fc, uint32_t* cc, BlockScratch3& scratch) {
  const uint64_t ma = __ballot(fa), mb = __ballot(fb), mc = __ballot(fc);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6u;
  auto prefix = [](uint64_t mask) { return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32u), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)); };
  if (lane == 0u) {
    scratch.wave_total[0][wave] = uint32_t(__popcll(ma));
    scratch.wave_total[1][wave] = uint32_t(__popcll(mb));
    scratch.wave_total[2][wave] = uint32_t(__popcll(mc));
  }
  __syncthreads();
  if (threadIdx.x < 3u) {  // one lane per counter: the three atomics are in flight together
    uint32_t total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
      total += scratch.wave_total[threadIdx.x][w];
    uint32_t* counter = (threadIdx.x == 0u) ? ca : ((threadIdx.x == 1u) ? cb : cc);
    scratch.base[threadIdx.x] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  Slots3 r = {scratch.base[0] + prefix(ma), scratch.base[1] + prefix(mb), scratch.base[2] + prefix(mc)};
#pragma unroll
  for (uint32_t w = 0; w < kBlockSize / 64u; ++w) {
    r.a += (w < wave) ? scratch.wave_total[0][w] : 0u;
    r.b += (w < wave) ? scratch.wave_total[1][w] : 0u;
    r.c += (w < wave) ? scratch.wave_total[2][w] : 0u;
  }
  __syncthreads();
  return r;
}

// Slot reservation policy of the step functions (dev_vcm_steps.h). The wavefront kernels reserve per workgroup; the
// tail kernels, whose lanes loop independently over the few surviving paths, reserve per lane.
struct BlockSlots {
  BlockScratch* scratch;
  BlockScratch3* scratch3;
  ETX_DEV uint32_t get(bool wanted, uint32_t* counter) const {
    return block_compact_slot(wanted, counter, *scratch);
  }
  ETX_DEV Slots3 get3(bool fa, uint32_t* ca, bool fb, uint32_t* cb, bool fc, uint32_t* cc) const {
    return block_compact_slot3(fa, ca, fb, cb, fc, cc, *scratch3);
  }
};
struct LaneSlots {
  ETX_DEV uint32_t get(bool wanted, uint32_t* counter) const {
    return wanted ? atomicAdd(counter, 1u) : 0u;
  }
  ETX_DEV Slots3 get3(bool fa, uint32_t* ca, bool fb, uint32_t* cb, bool fc, uint32_t* cc) const {
    return {get(fa, ca), get(fb, cb), get(fc, cc)};
  }
};

// Workgroup-uniform grid-stride loop: every thread of a workgroup runs the same number of iterations, threads beyond
// COUNT take part in the barriers of block_compact_slot with their predicate false.
#define ETX_BLOCK_LOOP(COUNT, INDEX)                                                                        \
  for (uint32_t block_base_ = blockIdx.x * blockDim.x, INDEX = block_base_ + threadIdx.x; block_base_ < (COUNT); \
       block_base_ += gridDim.x * blockDim.x, INDEX = block_base_ + threadIdx.x)

// Workgroup sum of a per-lane statistic into this workgroup's row of p.block_stats (no atomics).
ETX_DEV void block_stat_add(const Pipeline& p, uint32_t which, unsigned long long value, unsigned long long* scratch /* LDS, one word */) {
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1)
    value += __shfl_xor(value, d);
  if (threadIdx.x == 0u)
    *scratch = 0ull;
  __syncthreads();
  if (((threadIdx.x & 63u) == 0u) && (value != 0ull))
    atomicAdd(scratch, value);
  __syncthreads();
  if ((threadIdx.x == 0u) && (*scratch != 0ull))
    p.block_stats[min(blockIdx.x, kBlockStatRows - 1u) * kBlockStatCount + which] += *scratch;
  __syncthreads();
}

ETX_DEV void atomic_add_f3(float4* dst, const f3& v) {
  atomicAdd(&dst->x, v.x);
  atomicAdd(&dst->y, v.y);
  atomicAdd(&dst->z, v.z);
}

// Every film write goes through this guard: one non-finite contribution would poison its pixel for the rest of the run
// (all lanes, and after etx_hip_reduce_film all ranks, add into the same sums; Film::layer(Result) = max(0, ...) then shows
// it as black). Dropped contributions are counted (etx_hip_stats_t::nonfinite_dropped), never silent.
ETX_DEV bool film_value_ok(const Pipeline& p, const f3& v) {
  const bool ok = isfinite(v.x + v.y + v.z);
  if (ok == false)
    atomicAdd(p.counters + kCntNonFinite, 1u);
  return ok;
}
ETX_DEV void film_add(const Pipeline& p, float4* dst, const f3& v) {
  if (film_value_ok(p, v))
    atomic_add_f3(dst, v);
}

ETX_DEV bool is_zero(const f3& v) {  // SpectralResponse::is_zero, spectrum.hxx:317-319
  return (v.x <= kEpsilon) && (v.y <= kEpsilon) && (v.z <= kEpsilon);
}

// scene.hxx:228-248 random_continue (Russian roulette)
ETX_DEV bool random_continue(uint32_t path_length, uint32_t start_path_length, float eta_scale, Sampler& smp, f3& throughput) {
  float max_t = max_component(throughput);
  if (max_t == 0.0f)
    return false;
  if (path_length < start_path_length)
    return true;
  max_t *= sqr(eta_scale);
  if (valid_value(max_t) == false)
    return false;
  float q = fminf(0.95f, max_t);
  if ((q > 0.0f) && (smp.next() < q)) {
    throughput *= (1.0f / q);
    return true;
  }
  return false;
}

struct TraceCtx {  // what an inline traversal needs
  const DScene* scene;
  LaneStack stack;
  uint32_t alpha_seed;
};

ETX_DEV f3 trace_transmittance(TraceCtx& tc, const f3& p0, const f3& p1, uint32_t medium, float wavelength) {
  const DScene& s = *tc.scene;
  return bvh_transmittance(s, global_nodes(s), s.bvh_tris, s.bvh_root, tc.stack, p0, p1, medium, wavelength, tc.alpha_seed);
}

// A connection whose visibility is still unknown: the shade / connect kernels evaluate everything but the
// transmittance and queue the segment; k_trace_shadow (kernels_trace.hip) multiplies by the transmittance and adds
// the result to the film. `target`: film pixel index, bit 31 set = light image (splat), clear = camera image.
// The host's sample_blue_noise(pixel, scene.samples, sample, dimension 0..5) from the table of etx_hip_upload_bluenoise:
// 8 bytes per (pixel of the 128 x 128 tile, sample), float = (0.5 + byte) / 256 (thirdparty/bluenoise, wrap as there)
ETX_DEV void bluenoise_samples(const uint2* table, uint32_t px, uint32_t py, uint32_t sample, f2& d01, f2& d23, f2& d45) {
  const uint2 v = table[((py & 127u) * 128u + (px & 127u)) * 256u + (sample & 255u)];
  auto unpack = [](uint32_t word, uint32_t byte) { return (0.5f + float((word >> (8u * byte)) & 0xffu)) / 256.0f; };
  d01 = {unpack(v.x, 0), unpack(v.x, 1)};
  d23 = {unpack(v.x, 2), unpack(v.x, 3)};
  d45 = {unpack(v.y, 0), unpack(v.y, 1)};
}

struct ShadowRequest {
  f3 p0, p1, value;  // value: contribution as the film sees it (x spectral_film_weight in spectral mode)
  uint32_t medium, target;
  float wavelength;  // for the media along the segment
};

ETX_DEV void write_shadow(const Pipeline& p, uint32_t idx, const ShadowRequest& r) {
  if (idx >= p.shadow.capacity) {
    atomicOr(p.counters + kCntOverflow, kOverflowShadow);
    return;
  }
  p.shadow.p0_medium[idx] = mk4(r.p0, __uint_as_float(r.medium));
  p.shadow.p1_target[idx] = mk4(r.p1, __uint_as_float(r.target));
  p.shadow.value[idx] = mk4(r.value, r.wavelength);
}

// Endpoint connection request (pipeline.h EndpointQueue). `hit_or_pos`: (u, v, t, triangle) of a surface vertex or
// (position, kInvalid) of a medium vertex; `w_i`: direction of arrival; `throughput`: the vertex' throughput, already
// scaled by a subsurface walk when `exit_material` (the vertex is the exit point, shaded by scene.subsurface_exit_material).
// The request carries the connection's fixed randoms and a sampler stream of its own (the reference continues the
// path's stream through the stochastic evaluations; the step function keeps that stream for the continuation).
ETX_DEV void write_endpoint(const Pipeline& p, uint32_t idx, const float4& hit_or_pos, const f3& w_i, const f3& throughput, bool exit_material, float d_vcm, float d_vc, uint32_t depth,
  uint32_t medium, uint32_t id, float wavelength, const Sampler& smp) {
  if (idx >= p.endpoints.capacity) {
    atomicOr(p.counters + kCntOverflow, kOverflowEndpoints);
    return;
  }
  p.endpoints.hit[idx] = hit_or_pos;
  p.endpoints.wi_medium[idx] = mk4(w_i, __uint_as_float(medium));
  p.endpoints.thr_depth[idx] = mk4(throughput, __uint_as_float(depth | (exit_material ? kCvExitMaterialBit : 0u)));
  p.endpoints.mis_id[idx] = make_float4(d_vcm, d_vc, 0.0f, __uint_as_float(id));
  p.endpoints.rnd_seed[idx] = make_float4(smp.fixed_u, smp.fixed_v, smp.fixed_w, __uint_as_float(Sampler::random_seed(smp.seed, 0x454e4450u)));
  p.endpoints.wavelength[idx] = wavelength;
}

// vcm_shared.hxx:218-283 vcm_next_ray
// The material of a shading point as the simple shading group reads it (Lambert / translucent / mirror / delta conductor / boundary): the head of the
// reference's 200-byte record (reflectance, scattering, emission) and its scalar tail (class .. emission collimation) copied ONCE per shading point -
// adjacent fields read in one place become five wide loads - into a local record the step's helpers take instead of looking the table up again after
// every store (a step read the class, the scattering colour's indices and its colour four times over). The general groups keep the table reference
// (their out-of-line BSDFs take the material by index).
ETX_DEV void load_simple_material(const DScene& scene, uint32_t index, etx_abi_material& m) {
  const etx_abi_material& t = scene.materials[index];
  m.reflectance = t.reflectance, m.scattering = t.scattering, m.emission = t.emission;
  m.cls = t.cls, m.int_medium = t.int_medium, m.ext_medium = t.ext_medium, m.normal_image_index = t.normal_image_index, m.diffuse_variation = t.diffuse_variation;
  m.two_sided = t.two_sided, m.normal_scale = t.normal_scale, m.opacity = t.opacity, m.emission_collimation = t.emission_collimation;
  m.ext_ior = t.ext_ior, m.int_ior = t.int_ior;  // (the delta conductor of the simple group; they sit right before the class: one more wide load)
}

template <bool kSimple>
ETX_DEV bool vcm_next_ray(const DScene& scene, uint32_t path_source, PathState& st, const VcmParams& it, const Isect& isect, const BsdfData& bsdf_data, const BsdfSample& bs,
  bool subsurface_sample, const etx_abi_material& mat) {
  if (st.depth + 1 > scene.max_path_length)
    return false;
  if (bs.valid() == false)
    return false;
  const etx_abi_triangle& tri = scene.triangles[isect.tri];
  st.throughput *= bs.weight;
  if (path_source == kPathLight)
    st.throughput *= fix_shading_normal(isect.geo_n, isect.nrm, isect.w_i, bs.w_o);
  if (is_zero(st.throughput))
    return false;
  if (random_continue(st.depth, scene.random_path_termination, st.eta, st.sampler, st.throughput) == false)
    return false;
  if (bs.properties & kSampleMediumChanged)
    st.medium = bs.medium_index;
  float cos_theta_bsdf = fabsf(dot(isect.nrm, bs.w_o));
  if (bs.is_delta()) {
    st.d_vc *= cos_theta_bsdf;
    st.d_vm *= cos_theta_bsdf;
    st.d_vcm = 0.0f;
  } else {
    // vcm_shared.hxx:259-261: after a subsurface walk the reverse pdf is the cosine lobe of the exit point
    float rev_pdf = subsurface_sample ? (fabsf(dot(bsdf_data.w_i, isect.nrm)) / kPi) : bsdf_reverse_pdf_s<kSimple>(scene, bsdf_data, bs.w_o, mat, st.sampler);
    st.d_vc = (cos_theta_bsdf / bs.pdf) * (st.d_vc * rev_pdf + st.d_vcm + it.vm_weight);
    st.d_vm = (cos_theta_bsdf / bs.pdf) * (st.d_vm * rev_pdf + st.d_vcm * it.vc_weight + 1.0f);
    st.d_vcm = 1.0f / bs.pdf;
  }
  st.ray_d = bs.w_o;
  st.ray_o = shading_pos(scene, tri, isect.bc, bs.w_o);
  st.ray_tmax = kMaxFloat;
  st.ray_tmin = kRayEpsilon;
  st.eta *= bs.eta;
  st.depth += 1u;
  return true;
}

template <bool kSimple>
ETX_DEV bool vcm_next_ray(const DScene& scene, uint32_t path_source, PathState& st, const VcmParams& it, const Isect& isect, const BsdfData& bsdf_data, const BsdfSample& bs,
  bool subsurface_sample = false) {
  return vcm_next_ray<kSimple>(scene, path_source, st, it, isect, bsdf_data, bs, subsurface_sample, scene.materials[isect.material]);
}

// vcm_shared.hxx:436-449 vcm_handle_boundary_bsdf
ETX_DEV bool vcm_handle_boundary(const DScene& scene, const Isect& isect, PathState& st, const etx_abi_material& mat) {
  if (mat.cls != ETX_MAT_BOUNDARY)
    return false;
  uint32_t new_medium = (dot(isect.geo_n, st.ray_d) < 0.0f) ? mat.int_medium : mat.ext_medium;
  st.path_distance += isect.t;
  st.medium = new_medium;
  st.ray_o = shading_pos(scene, scene.triangles[isect.tri], isect.bc, st.ray_d);
  st.ray_tmax = kMaxFloat;
  st.ray_tmin = kRayEpsilon;
  return true;
}
ETX_DEV bool vcm_handle_boundary(const DScene& scene, const Isect& isect, PathState& st) {
  return vcm_handle_boundary(scene, isect, st, scene.materials[isect.material]);
}

// vcm_shared.hxx:463-535 vcm_connect_to_camera, minus the transmittance: queues the segment, the splat value
// (vcm_cpu.cxx:148-153) is completed by k_trace_shadow.
// Returns true when `out` holds a request for the shadow queue.
template <bool kSimple>
ETX_DEV bool vcm_connect_to_camera(const DScene& scene, const VcmParams& it, bool camera_at_medium, const Isect* isect, const f3& medium_pos, PathState& st, ShadowRequest& out,
  const etx_abi_material& mat) {  // mat: the vertex' material (unused at a medium vertex)
  if ((opt_connect_to_camera(it) == false) || (st.depth + 2 > scene.max_path_length) || (st.depth + 2 < scene.min_path_length))
    return false;
  f3 sample_pos = camera_at_medium ? medium_pos : isect->pos;
  CameraSample cs = sample_film(scene, st.sampler, sample_pos);
  if (cs.pdf_dir <= 0.0f)
    return false;
  f3 direction = cs.position - sample_pos;
  float dist2 = dot(direction, direction);
  if (dist2 <= kEpsilon)
    return false;
  f3 w_o = normalize(direction);
  f3 scatter = mk3(0.0f);
  float reverse_pdf = 0.0f;
  f3 origin = sample_pos;
  if (camera_at_medium == false) {
    BsdfData data = make_bsdf_data(*isect, isect->w_i, st.medium, kPathLight, st.wavelength);
    BsdfEval eval = bsdf_evaluate_s<kSimple>(scene, data, w_o, mat, st.sampler);
    if (eval.valid() == false)
      return false;
    scatter = eval.bsdf;
    reverse_pdf = bsdf_reverse_pdf_s<kSimple>(scene, data, w_o, mat, st.sampler);
    origin = shading_pos(scene, scene.triangles[isect->tri], isect->bc, w_o);
  } else {
    const DMedium& medium = scene.mediums[st.medium];
    float p = phase_function(st.ray_d, w_o, medium.g);
    if (p <= 0.0f)
      return false;
    scatter = mk3(p);
    reverse_pdf = phase_function(w_o, st.ray_d, medium.g);
  }
  float len = length(cs.position - origin);
  float cos_t = fabsf(dot(cs.direction, scene.camera.direction));
  f3 clip_pos = origin + cs.direction * fmaxf(0.0f, len - scene.camera.clip_near / cos_t);
  float camera_pdf = cs.pdf_dir_out * (camera_at_medium ? 1.0f : fabsf(dot(isect->nrm, w_o))) / dist2;
  float vmW_cam = camera_at_medium ? 0.0f : it.vm_weight;
  float w_light = camera_pdf * (vmW_cam + st.d_vcm + st.d_vc * reverse_pdf);
  float weight = opt_enable_mis(it) ? (1.0f / (1.0f + w_light)) : 1.0f;
  if (camera_at_medium == false)
    weight *= fix_shading_normal(isect->geo_n, isect->nrm, isect->w_i, w_o);
  // film.cxx:147-171 atomic_add_light_iteration: NDC -> pixel, y flip
  uint32_t x = static_cast<uint32_t>((cs.uv.x * 0.5f + 0.5f) * float(it.film_w));
  uint32_t y = static_cast<uint32_t>((cs.uv.y * 0.5f + 0.5f) * float(it.film_h));
  if ((x >= it.film_w) || (y >= it.film_h))
    return false;
  out = {origin, clip_pos, scatter * st.throughput * (cs.weight * weight) * spectral_film_weight(scene, st.wavelength), st.medium,
    kShadowTargetLight | (x + (it.film_h - 1u - y) * it.film_w), st.wavelength};
  return true;
}

// vcm_shared.hxx:285-308 vcm_get_radiance (direct emitter hit from the camera sub path)
ETX_DEV f3 vcm_get_radiance(const DScene& scene, const etx_abi_emitter& emitter, const PathState& st, const VcmParams& it, const Isect& isect) {
  float pdf_area = 0.0f, pdf_dir = 0.0f, pdf_dir_out = 0.0f;
  EmitterRadianceQuery q;
  q.source_position = st.ray_o;
  q.target_position = isect.pos;
  q.direction = st.ray_d;
  q.uv = isect.tex;
  q.directly_visible = st.depth == 1;
  f3 radiance = emitter_get_radiance(scene, emitter, q, pdf_area, pdf_dir, pdf_dir_out, st.wavelength);
  if (pdf_dir <= kEpsilon)
    return mk3(0.0f);
  float pdf_sample = emitter_discrete_pdf(scene, emitter);
  float w_camera = st.d_vcm * pdf_area * pdf_sample + st.d_vc * (pdf_dir_out * pdf_sample);
  float weight = (opt_enable_mis(it) && (st.depth > 1)) ? (1.0f / (1.0f + w_camera)) : 1.0f;
  return weight * (st.throughput * radiance);
}

// vcm_shared.hxx:537-587 vcm_cam_handle_miss
ETX_DEV f3 vcm_cam_handle_miss(const DScene& scene, const VcmParams& it, PathState& st) {
  if (opt_direct_hit(it) == false)
    return mk3(0.0f);
  if (st.path_distance > 0.0f) {
    st.d_vcm *= sqr(st.path_distance);
    st.path_distance = 0.0f;
  }
  f3 accumulated = mk3(0.0f);
  float sum_pdf_dir_out = 0.0f, sum_pdf_dir = 0.0f;
  for (uint32_t ie = 0; ie < scene.env_count; ++ie) {
    const etx_abi_emitter& em = scene.emitters[scene.env_emitters[ie]];
    EmitterRadianceQuery q;
    q.source_position = q.target_position = mk3(0.0f);
    q.direction = st.ray_d;
    q.uv = {0.0f, 0.0f};
    q.directly_visible = st.depth <= 1;
    float pdf_area = 0.0f, pdf_dir = 0.0f, pdf_dir_out = 0.0f;
    f3 value = emitter_get_radiance(scene, em, q, pdf_area, pdf_dir, pdf_dir_out, st.wavelength);
    if (pdf_dir > kEpsilon) {
      float pdf_discrete = emitter_discrete_pdf(scene, em);
      sum_pdf_dir_out += pdf_dir_out * pdf_discrete;
      sum_pdf_dir += pdf_dir * pdf_discrete;
      accumulated += value;
    }
  }
  if (max_component(accumulated) > kEpsilon) {
    float inv_count = (scene.env_count > 0u) ? (1.0f / float(scene.env_count)) : 0.0f;
    sum_pdf_dir *= inv_count;
    sum_pdf_dir_out *= inv_count;
    float w_camera_sum = st.d_vcm * sum_pdf_dir + st.d_vc * sum_pdf_dir_out;
    float weight = (opt_enable_mis(it) && (st.depth > 1)) ? (1.0f / (1.0f + w_camera_sum)) : 1.0f;
    return st.throughput * accumulated * weight;
  }
  return mk3(0.0f);
}

// vcm_shared.hxx:608-671 vcm_connect_to_light (NEE) minus the transmittance (queued for k_trace_shadow).
// sampler.fixed_* hold (rnd_connection.xy, rnd_support.y). `film_target` = film index of the path's pixel.
// Returns true when `out` holds a request for the shadow queue.
template <bool kSimple>
ETX_DEV bool vcm_connect_to_light(const DScene& scene, const VcmParams& it, bool camera_at_medium, const Isect* isect, const f3& medium_pos, PathState& st, uint32_t film_target, ShadowRequest& out,
  const etx_abi_material& mat) {  // mat: the vertex' material (unused at a medium vertex)
  if ((opt_connect_to_light(it) == false) || (st.depth + 1 > scene.max_path_length) || (st.depth + 1 < scene.min_path_length))
    return false;
  f3 sample_pos = camera_at_medium ? medium_pos : isect->pos;
  uint32_t emitter_index = sample_emitter_index(scene, st.sampler.fixed_w);
  EmitterSample es = sample_emitter(scene, emitter_index, f2{st.sampler.fixed_u, st.sampler.fixed_v}, sample_pos, st.wavelength);
  if (es.pdf_dir <= 0.0f)
    return false;
  f3 w_o = es.direction;
  f3 scatter = mk3(0.0f);
  float reverse_pdf = 0.0f;
  f3 origin = sample_pos;
  float camera_factor = 1.0f;
  float conn_pdf = 0.0f;
  if (camera_at_medium) {
    const DMedium& medium = scene.mediums[st.medium];
    float p = phase_function(st.ray_d, w_o, medium.g);
    if (p <= 0.0f)
      return false;
    scatter = mk3(p);
    reverse_pdf = phase_function(w_o, st.ray_d, medium.g);
  } else {
    BsdfData data = make_bsdf_data(*isect, isect->w_i, st.medium, kPathCamera, st.wavelength);
    BsdfEval eval = bsdf_evaluate_s<kSimple>(scene, data, w_o, mat, st.sampler);
    if (eval.valid() == false)
      return false;
    scatter = eval.bsdf;
    reverse_pdf = bsdf_reverse_pdf_s<kSimple>(scene, data, w_o, mat, st.sampler);
    const etx_abi_triangle& tri = scene.triangles[isect->tri];
    origin = shading_pos(scene, tri, isect->bc, normalize(es.origin - isect->pos));
    camera_factor = fabsf(dot(w_o, isect->geo_n));
    conn_pdf = bsdf_pdf_s<kSimple>(scene, data, w_o, mat, st.sampler);
  }
  float l_dot_e = fabsf(dot(es.direction, es.normal));
  float w_light = 0.0f;
  if (es.is_delta == false) {
    // vcm_shared.hxx:655: the medium branch divides the scalar phase value (SpectralResponse::value)
    w_light = (camera_at_medium ? scatter.x : conn_pdf) / (es.pdf_dir * es.pdf_sample);
  }
  float vmW_nee = camera_at_medium ? 0.0f : it.vm_weight;
  float w_camera = (es.pdf_dir_out * camera_factor) / (es.pdf_dir * l_dot_e) * (vmW_nee + st.d_vcm + st.d_vc * reverse_pdf);
  float weight = opt_enable_mis(it) ? 1.0f / (1.0f + w_light + w_camera) : 1.0f;
  out = {origin, es.origin, st.throughput * scatter * es.value * (weight / (es.pdf_dir * es.pdf_sample)) * spectral_film_weight(scene, st.wavelength), st.medium, film_target,
    st.wavelength};
  return true;
}

// Pairs of the bounce for the connection kernels. k_expand_pairs reserves a camera vertex' whole run of pairs and writes nothing of a run that
// does not fit the buffer: the list then has HOLES (whatever the memory held) below its end. The bounce has overflowed and its iteration will
// be rendered again with a larger buffer (host_api.cpp execute_iteration) - nothing of it is evaluated.
ETX_DEV uint32_t pair_list_count(const Pipeline& p) {
  if (p.counters[kCntOverflow] & kOverflowPairs)
    return 0u;
  return min(p.counters[kCntPairs], p.pair_capacity);
}

struct LightVertex {  // VCMLightVertex, vcm_shared.hxx:154-197
  f3 pos, w_i, throughput, nrm;
  float d_vcm, d_vc, d_vm;
  float bc_u, bc_v;
  uint32_t tri, path_length, index_in_path, medium;
  ETX_DEV bool is_medium() const {
    return tri == kInvalid;
  }
};

ETX_DEV LightVertex load_light_vertex(const LightVertexPool& lv, uint32_t i) {
  float4 a = lv.pos_dvcm(i), b = lv.wi_dvc(i), c = lv.thr_dvm(i), d = lv.nrm_tri(i), e = lv.bc_len_med(i);
  LightVertex v;
  v.pos = {a.x, a.y, a.z}, v.d_vcm = a.w;
  v.w_i = {b.x, b.y, b.z}, v.d_vc = b.w;
  v.throughput = {c.x, c.y, c.z}, v.d_vm = c.w;
  v.nrm = {d.x, d.y, d.z}, v.tri = __float_as_uint(d.w);
  v.bc_u = e.x, v.bc_v = e.y, v.path_length = __float_as_uint(e.z) & 0xffffu, v.index_in_path = __float_as_uint(e.z) >> 16u, v.medium = __float_as_uint(e.w);
  return v;
}

// Surface BSDF access for the connection / merge kernels: kDiffuseOnly instantiations are launched for vertices whose
// materials are all Material::Class::Diffuse (the common case; keeps the Heitz random walk out of the register budget),
// the generic instantiation dispatches over every implemented class.
template <bool kDiffuseOnly>
ETX_DEV BsdfEval bsdf_evaluate_t(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m, Sampler& smp) {
  if (kDiffuseOnly)
    return diffuse_evaluate(s, d, w_o, m);
  return bsdf_evaluate_general(s, d, w_o, uint32_t(&m - s.materials), smp);
}
template <bool kDiffuseOnly>
ETX_DEV float bsdf_reverse_pdf_t(const DScene& s, const BsdfData& d, const f3& w_o, const etx_abi_material& m, Sampler& smp) {
  if (kDiffuseOnly) {
    BsdfData r = d;
    r.w_i = -w_o;
    return diffuse_pdf(r, -d.w_i);
  }
  return bsdf_reverse_pdf_s<false>(s, d, w_o, m, smp);
}

// vcm_shared.hxx:673-763 vcm_connect_to_light_vertex (surface / medium on either side)
template <bool kDiffuseOnly>
ETX_DEV bool vcm_connect_to_light_vertex(const DScene& scene, const PathState& st, const LightVertex& lv, const VcmParams& it, bool camera_at_medium, const Isect* cam, const f3& medium_pos,
  Sampler& smp, f3& target_position, f3& value) {
  Vtx light_v;
  f3 light_geo_n = mk3(0.0f);  // the geometric normal and the material index come with the triangle's rows
  uint32_t light_material = 0u;
  if (lv.is_medium() == false) {
    if (kDiffuseOnly && (scene.textured_materials == 0u)) {
      // no normal map, no texture anywhere: the interpolated vertex the reference rebuilds here (lerp_vertex, vcm_shared.hxx:770-772) is the
      // position and shading normal the record already holds, and a Lambert BSDF reads nothing else of it - one row instead of seven
      const float4 row = scene.tri_shade[size_t(lv.tri) * kTriShadeStride + 6u];  // geometric normal, material index
      light_v.pos = lv.pos, light_v.nrm = lv.nrm;
      light_v.tan = light_v.btn = mk3(0.0f), light_v.tex = {0.0f, 0.0f};
      light_geo_n = xyz(row), light_material = __float_as_uint(row.w);
    } else {
      const TriPoint point = lerp_tri_point(scene, scene.triangles[lv.tri], barycentrics(lv.bc_u, lv.bc_v));
      light_v = point.v, light_geo_n = point.tv.geo_n, light_material = point.material;
    }
  }
  target_position = lv.is_medium() ? lv.pos : light_v.pos;
  f3 w_o = target_position - (camera_at_medium ? medium_pos : cam->pos);
  float distance_squared = dot(w_o, w_o);
  if (distance_squared <= kEpsilon)
    return false;
  w_o = w_o / sqrtf(distance_squared);
  float w_dot_l = lv.is_medium() ? 1.0f : -dot(light_v.nrm, w_o);

  float camera_area_pdf = 0.0f, camera_rev_pdf = 0.0f;
  f3 camera_scatter = mk3(0.0f);
  if (camera_at_medium) {
    const DMedium& medium = scene.mediums[st.medium];
    float p = phase_function(st.ray_d, w_o, medium.g);
    if (p <= 0.0f)
      return false;
    camera_area_pdf = p * fabsf(w_dot_l) / distance_squared;
    camera_rev_pdf = phase_function(w_o, st.ray_d, medium.g);
    camera_scatter = mk3(p);
  } else {
    const etx_abi_material& mat = scene.materials[cam->material];
    BsdfData camera_data = make_bsdf_data(*cam, cam->w_i, st.medium, kPathCamera, st.wavelength);
    BsdfEval camera_bsdf = bsdf_evaluate_t<kDiffuseOnly>(scene, camera_data, w_o, mat, smp);
    if (camera_bsdf.valid() == false)
      return false;
    camera_area_pdf = camera_bsdf.pdf * fabsf(w_dot_l) / distance_squared;
    camera_rev_pdf = bsdf_reverse_pdf_t<kDiffuseOnly>(scene, camera_data, w_o, mat, smp);
    camera_scatter = camera_bsdf.bsdf;
  }

  float light_area_pdf = 0.0f, light_rev_pdf = 0.0f;
  f3 light_scatter = mk3(0.0f);
  if (lv.is_medium()) {
    const DMedium& med = scene.mediums[lv.medium];
    float p = phase_function(lv.w_i, -w_o, med.g);
    if (p <= 0.0f)
      return false;
    light_area_pdf = p * (camera_at_medium ? 1.0f : fabsf(dot(cam->nrm, w_o))) / distance_squared;
    light_rev_pdf = phase_function(-w_o, lv.w_i, med.g);
    light_scatter = mk3(p);
  } else {
    const etx_abi_material& light_mat = scene.materials[light_material];
    BsdfData light_data = make_bsdf_data(light_v, lv.w_i, camera_at_medium ? lv.medium : st.medium, kPathLight, st.wavelength);
    BsdfEval light_bsdf = bsdf_evaluate_t<kDiffuseOnly>(scene, light_data, -w_o, light_mat, smp);
    if (light_bsdf.valid() == false)
      return false;
    light_area_pdf = camera_at_medium ? (light_bsdf.pdf / distance_squared) : (light_bsdf.pdf * fabsf(dot(cam->nrm, w_o)) / distance_squared);
    light_rev_pdf = bsdf_reverse_pdf_t<kDiffuseOnly>(scene, light_data, -w_o, light_mat, smp);
    light_scatter = light_bsdf.bsdf * fix_shading_normal(light_geo_n, light_data.nrm, light_data.w_i, -w_o);
  }

  float vmW_pair = (camera_at_medium || lv.is_medium()) ? 0.0f : it.vm_weight;
  float w_light = camera_area_pdf * (vmW_pair + lv.d_vcm + lv.d_vc * light_rev_pdf);
  float w_camera = light_area_pdf * (vmW_pair + st.d_vcm + st.d_vc * camera_rev_pdf);
  float weight = opt_enable_mis(it) ? 1.0f / (1.0f + w_light + w_camera) : 1.0f;
  value = (camera_scatter * st.throughput) * (light_scatter * lv.throughput) * (weight / distance_squared);
  return true;
}

ETX_DEV uint32_t grid_cell_index(int32_t x, int32_t y, int32_t z, uint32_t mask) {  // vcm_shared.hxx:820-822
  return ((uint32_t(x) * 73856093u) ^ (uint32_t(y) * 19349663u) ^ (uint32_t(z) * 83492791u)) & mask;
}

ETX_DEV uint32_t grid_position_to_index(const GridParams& g, const f3& pos) {  // vcm_shared.hxx:824-827
  f3 m = (pos - g.bbox_min) / g.cell_size;
  return grid_cell_index(int32_t(floorf(m.x)), int32_t(floorf(m.y)), int32_t(floorf(m.z)), g.hash_mask);
}


// film.cxx:189: pixel (x, y) is stored in row (H - 1 - y)
ETX_DEV uint32_t film_index(const VcmParams& it, uint32_t pixel_id) {
  uint32_t px = pixel_id % it.film_w, py = pixel_id / it.film_w;
  return px + (it.film_h - 1u - py) * it.film_w;
}

// A connectible camera vertex as k_camera_shade stored it (CameraVertexPool): the state BEFORE vcm_next_ray, after
// vcm_update_camera_vcm.
struct CameraVertex {
  PathState st;   // throughput, d_vcm/d_vc/d_vm, depth, medium, ray_d (= w_i), id (pixel)
  Isect isect;
  f3 medium_pos;
  bool at_medium;
};

ETX_DEV CameraVertex load_camera_vertex(const Pipeline& p, const DScene& scene, uint32_t i) {
  CameraVertex cv;
  float4 h = p.cv.hit[i], w = p.cv.wi_medium[i], t = p.cv.thr_depth[i], m = p.cv.mis_pixel[i];
  cv.st.ray_d = {w.x, w.y, w.z};
  cv.st.medium = __float_as_uint(w.w);
  cv.st.throughput = {t.x, t.y, t.z};
  const uint32_t depth_bits = __float_as_uint(t.w);
  cv.st.depth = depth_bits & 0x7fffffffu;
  cv.st.d_vcm = m.x, cv.st.d_vc = m.y, cv.st.d_vm = m.z;
  cv.st.id = __float_as_uint(m.w);
  cv.st.sampler.seed = p.cv.seed[i];
  cv.st.wavelength = p.cv.wavelength[i];
  cv.st.sampler.fixed_u = cv.st.sampler.fixed_v = cv.st.sampler.fixed_w = 0.0f;
  cv.st.eta = 1.0f, cv.st.path_distance = 0.0f, cv.st.flags = 0u;
  cv.st.ray_o = mk3(0.0f), cv.st.ray_tmin = 0.0f, cv.st.ray_tmax = 0.0f;
  uint32_t tri = __float_as_uint(h.w);
  cv.at_medium = tri == kInvalid;
  cv.medium_pos = {h.x, h.y, h.z};
  if (cv.at_medium == false)
    cv.isect = make_intersection(scene, cv.st.ray_d, h.x, h.y, h.z, tri);
  if ((cv.at_medium == false) && (depth_bits & kCvExitMaterialBit))
    cv.isect.material = scene.subsurface_exit_material;  // exit point of a subsurface walk (vcm_shared.hxx:1037-1038)
  return cv;
}

// An endpoint connection request as the step functions wrote it (EndpointQueue): the vertex with the state
// vcm_connect_to_camera / vcm_connect_to_light read.
ETX_DEV CameraVertex load_endpoint(const Pipeline& p, const DScene& scene, uint32_t i) {
  CameraVertex cv;
  const float4 h = p.endpoints.hit[i], w = p.endpoints.wi_medium[i], t = p.endpoints.thr_depth[i], m = p.endpoints.mis_id[i], r = p.endpoints.rnd_seed[i];
  cv.st.ray_d = {w.x, w.y, w.z};
  cv.st.medium = __float_as_uint(w.w);
  cv.st.throughput = {t.x, t.y, t.z};
  const uint32_t depth_bits = __float_as_uint(t.w);
  cv.st.depth = depth_bits & 0x7fffffffu;
  cv.st.d_vcm = m.x, cv.st.d_vc = m.y, cv.st.d_vm = 0.0f;
  cv.st.id = __float_as_uint(m.w);
  cv.st.sampler.seed = __float_as_uint(r.w);
  cv.st.sampler.fixed_u = r.x, cv.st.sampler.fixed_v = r.y, cv.st.sampler.fixed_w = r.z;
  cv.st.wavelength = p.endpoints.wavelength[i];
  cv.st.eta = 1.0f, cv.st.path_distance = 0.0f, cv.st.flags = 0u;
  cv.st.ray_o = mk3(0.0f), cv.st.ray_tmin = 0.0f, cv.st.ray_tmax = 0.0f;
  const uint32_t tri = __float_as_uint(h.w);
  cv.at_medium = tri == kInvalid;
  cv.medium_pos = {h.x, h.y, h.z};
  if (cv.at_medium == false)
    cv.isect = make_intersection(scene, cv.st.ray_d, h.x, h.y, h.z, tri);
  if ((cv.at_medium == false) && (depth_bits & kCvExitMaterialBit))
    cv.isect.material = scene.subsurface_exit_material;
  return cv;
}

}  // namespace etxd
