// kernels_bdpt.hip - bidirectional path tracing (CPUBidirectional, sources/etx/rt/integrators/bidirectional.cxx) on the
// wavefront pipeline. Kernel sequence of one iteration (host_api.cpp render_bdpt_iteration):
//   k_bdpt_light_generate                                   build_emitter_path, start            (:955-1010)
//   loop: k_trace_closest ; k_bdpt_light_shade ;            build_path, Light mode               (:820-898, handle_surface :572-702,
//         k_bdpt_connect_camera ; k_trace_shadow                                                   handle_medium :533-570, connect_light_to_camera :1380-1428)
//   k_bdpt_camera_generate                                  build_camera_path, start             (:900-953)
//   loop: k_trace_closest ; k_bdpt_camera_shade ;           build_path, Camera mode + direct hits (:1235-1340)
//         k_bdpt_connect_light ;                            connect_camera_to_light              (:1342-1378)
//         k_bdpt_expand_pairs ; k_bdpt_connect_pairs ;      connect_camera_to_light_path         (:438-497, MIS :1184-1209)
//         k_trace_shadow
//   k_vcm_commit                                            Film::commit_light_iteration + the iteration's camera estimate
// dev_bdpt.h explains the data layout. Scenes of simple BSDF classes run inline instantiations, mixed scenes split their items by class (kPartSimple / kPartGeneral).
#include "kernels.h"
#include "dev_bdpt.h"

namespace etxd {

static uint32_t grid_for(uint32_t capacity) {
  return min(kPersistentBlocks, (capacity + kBlockSize - 1) / kBlockSize);
}

ETX_DEV uint32_t bdpt_mode(const VcmParams& it) {
  return it.kernel;  // the host passes CPUBidirectionalImpl::Mode here (VcmParams is shared with VCM)
}

// Film::sample, film.cxx:137-145 (the pixel filter of PT / BDPT; the first iteration uses PixelFilter::empty())
ETX_DEV f2 bdpt_film_sample(const DScene& scene, bool filtered, uint32_t px, uint32_t py, const VcmParams& it, const f2 rnd) {
  f2 jitter = {rnd.x * 2.0f - 1.0f, rnd.y * 2.0f - 1.0f};
  float radius = 0.0f;
  if (filtered) {
    radius = scene.pixel_sampler_radius;
    if (scene.pixel_sampler_image != kInvalid) {
      float pdf = 0.0f;
      float4 eval;
      const f2 uv = image_sample(scene.images[scene.pixel_sampler_image], rnd, pdf, eval);
      jitter = {uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f};
    }
  }
  return {(float(px) + 0.5f + radius * jitter.x) / float(it.film_w) * 2.0f - 1.0f, (float(py) + 0.5f + radius * jitter.y) / float(it.film_h) * 2.0f - 1.0f};
}

// bsdf::albedo (scene_bsdf.hxx:95-107), as kernels_pt.hip
ETX_DEV f3 bdpt_albedo(const DScene& scene, const etx_abi_material& mat, const f2 tex, float wavelength) {
  switch (mat.cls) {
    case ETX_MAT_CONDUCTOR:
      return apply_image(scene, mat.reflectance, tex, nullptr, wavelength);
    case ETX_MAT_MIRROR:
    case ETX_MAT_BOUNDARY:
      return mk3(1.0f);
    case ETX_MAT_VOID:
      return mk3(0.0f);
    default:
      return apply_image(scene, mat.scattering, tex, nullptr, wavelength);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// build_emitter_path, bidirectional.cxx:955-1010
__global__ __launch_bounds__(kBlockSize) void k_bdpt_light_generate(Pipeline p, VcmParams it) {
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  ETX_BLOCK_LOOP(it.path_count, k) {
    bool valid = false;
    BdptState st = {};
    if (k < it.path_count) {
      const uint32_t i = path_pixel(it, k);  // the emitter path of pixel i (:377-379)
      st.sampler.init(i, it.iteration);
      st.id = i;
      st.wavelength = scene.spectral ? spectral_sample_wavelength(st.sampler.next()) : 0.0f;
      p.path_wavelength[i] = st.wavelength;
      const EmitterSample es = sample_emission(scene, st.sampler, st.wavelength);
      if ((es.pdf_area != 0.0f) && (es.pdf_dir != 0.0f) && (is_zero(es.value) == false)) {
        st.throughput = es.value * (dot(es.direction, es.normal) / (es.pdf_dir * es.pdf_area * es.pdf_sample));
        st.ray_o = offset_ray(es.origin, es.normal);
        st.ray_d = es.direction;
        st.ray_tmin = kRayEpsilon, st.ray_tmax = kMaxFloat;
        st.eta = 1.0f;
        st.pdf_dir = es.pdf_dir;
        st.mis_history = (bdpt_mode(it) == kBdptFast) ? 1.0f : 0.0f;  // :980-984
        st.aux = es.pdf_area;
        st.path_size = 1u;
        st.medium = es.medium_index;
        st.flags = kBpFirst | (es.is_distant ? kBpDistantEmitter : 0u) | (es.emitter_index << kBpEmitterShift);
        // the emitter vertex itself = emitter_path[0]
        st.prev.pos = es.origin, st.prev.nrm = es.normal;
        st.prev.from_prev = es.pdf_area * es.pdf_sample;
        st.prev.flags = kBvEmitter | kBvConnectible | (es.is_delta ? 0u : kBvMisConnectible) | ((es.triangle_index != kInvalid) ? kBvSurface : 0u);
        st.prev.tri = es.triangle_index;
        valid = true;
      }
    }
    const uint32_t slot = block_compact_slot(valid, p.counters + kCntActiveA, s_scratch);
    if (valid)
      bdpt_store(p.paths[0], slot, st, st.prev.tri);  // until the emitter vertex is in the pool the slot field carries its triangle
  }
}

// Subsurface walks (build_path: subsurface_material; subsurface_step, bidirectional.cxx:746-818). A path that enters a subsurface
// object leaves the wavefront: the shade kernel writes its state to the WALK QUEUE (Pipeline::walk) instead of the next round's
// path set, and k_bdpt_walk - persistent wavefronts that take entries from that queue whenever enough of their lanes are idle -
// runs the walk: free flight through the walk's medium, a material-filtered closest-hit query in place of the ray queue, one path
// vertex per scattering event, until the path has left the object; then it joins the next round's path set. Every sub-step goes
// through the same vertex code as a queue segment (bdpt_light_step / bdpt_camera_step).
struct BdptWalk {
  uint32_t material;  // kInvalid: not inside an object
  uint32_t medium;    // DScene::material_sss_medium of that material
  uint32_t events;
};

// subsurface_step's free flight: returns false when the path ends (pdf zero). found = the object's surface was reached (h filled).
template <class Nodes>
ETX_DEV bool bdpt_walk_flight(const DScene& scene, const Nodes& nodes, const LaneStack& stack, const BdptWalk& walk, BdptState& st, float4& h, MediumSample& ms) {
  const DMedium& wm = scene.mediums[walk.medium];
  f3 absorption, scattering;
  medium_coefficients(scene, wm, st.wavelength, absorption, scattering);
  const f3 extinction = scattering + absorption;
  const f3 albedo = {extinction.x > 0.0f ? scattering.x / extinction.x : 0.0f, extinction.y > 0.0f ? scattering.y / extinction.y : 0.0f, extinction.z > 0.0f ? scattering.z / extinction.z : 0.0f};
  f3 pdf = mk3(0.0f);
  float max_t = 0.0f;
  while (max_t < kRayEpsilon) {
    const uint32_t channel = sample_spectrum_component(albedo, st.throughput, st.sampler.next(), pdf);
    const float sample_t = channel == 0 ? extinction.x : (channel == 1 ? extinction.y : extinction.z);
    max_t = (sample_t > 0.0f) ? -logf(1.0f - st.sampler.next()) / sample_t : kMaxFloat;
  }
  uint32_t alpha_seed = st.sampler.seed ^ 0x62777373u;
  const Hit hit = bvh_closest(scene, nodes, scene.bvh_tris, scene.bvh_root, stack, RayQ{st.ray_o, st.ray_tmin, st.ray_d, max_t}, alpha_seed, nullptr, walk.material);
  const bool found = hit.tri != kInvalid;
  if (found)
    max_t = hit.t;
  const f3 tr = {expf(-max_t * extinction.x), expf(-max_t * extinction.y), expf(-max_t * extinction.z)};
  pdf = found ? pdf * tr : pdf * tr * extinction;
  if ((pdf.x == 0.0f) && (pdf.y == 0.0f) && (pdf.z == 0.0f))
    return false;
  st.throughput *= (found ? tr : tr * scattering) / (pdf.x + pdf.y + pdf.z);
  h = make_float4(hit.u, hit.v, hit.t, __uint_as_float(hit.tri));
  ms.weight = mk3(1.0f);
  ms.pos = st.ray_o + st.ray_d * max_t;
  ms.sampled_medium_t = found ? 0.0f : max_t;
  return true;
}

// handle_surface :610-633: a diffuse reflection off a subsurface material enters the object instead (the vertex gets the scatter
// material and a cosine lobe into the object or the incoming direction as its sampled direction). Three media leave here:
// walk_medium, the one the walk flies through (subsurface_step derives it from the object's material, :757-771); vertex_medium, the
// instance the entry VERTEX keeps for the transmittance of its connections - the interior medium, or, without one, the instance
// :632 derives from the SCATTER material (material_index was swapped at :630), an entry host_scene.cpp builds for that material;
// and the path's medium index, which is the interior medium or none (:670 payload.medium_index = medium_instance.index).
// A slot for every lane that calls this (any subset of a wavefront: the callers sit in divergent shading code): one atomic per call site and wavefront.
ETX_DEV uint32_t divergent_slot(uint32_t* counter) {
  const unsigned long long mask = __ballot(true);
  const uint32_t lane = __lane_id();
  const uint32_t leader = uint32_t(__ffsll((long long)mask)) - 1u;
  uint32_t base = 0u;
  if (lane == leader)
    base = atomicAdd(counter, uint32_t(__popcll(mask)));
  base = __shfl(base, int(leader));
  return base + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
}

// The walk medium of a subsurface material WITHOUT an interior medium whose colour or distances are textured (subsurface_step, bidirectional.cxx:757-765:
// apply_image at the entry point's texture coordinate, subsurface::remap per channel): a row of its own behind the lane's copy of the medium table. RGB mode:
// three channels; spectral mode: the path's wavelength, replicated (kMediumStoredCoefficients makes medium_coefficients take the row as it is).
ETX_DEV uint32_t bdpt_derive_walk_medium(const DScene& scene, const etx_abi_material& mat, const Isect& isect, float wavelength, uint32_t fallback) {
  const f3 color = apply_image(scene, mat.scattering, isect.tex, nullptr, wavelength);
  const etx_abi_spectral_image distances_image = {mat.subsurface.spectrum_index, mat.subsurface.image_index};
  const f3 distances = apply_image(scene, distances_image, isect.tex, nullptr, wavelength);
  f3 albedo, extinction, scattering;
  sss_remap_channel(color.x, distances.x, albedo.x, extinction.x, scattering.x);
  sss_remap_channel(color.y, distances.y, albedo.y, extinction.y, scattering.y);
  sss_remap_channel(color.z, distances.z, albedo.z, extinction.z, scattering.z);
  const uint32_t slot = divergent_slot(scene.lane_counters + kCntDynMedium);
  if (slot >= scene.dyn_medium_capacity) {  // the iteration is discarded, the pools (and this table with them) grow, it is rendered again
    atomicOr(scene.lane_counters + kCntOverflow, kOverflowLightVertices);
    return fallback;
  }
  const uint32_t index = scene.dyn_medium_first + slot;
  float4* row = reinterpret_cast<float4*>(const_cast<DMedium*>(scene.mediums) + index);
  const f3 absorption = extinction - scattering;
  static_assert(sizeof(DMedium) == 8u * sizeof(float4), "a medium row is written as eight float4");
  static_assert((offsetof(DMedium, density) == 48u) && (offsetof(DMedium, absorption_index) == 80u) && (offsetof(DMedium, derived_color) == 112u), "DMedium layout");
  row[0] = mk4(absorption, 0.0f);                                    // absorption, g
  row[1] = mk4(scattering, __uint_as_float(0u));                     // scattering, cls | explicit_connections << 16: homogeneous, no explicit connections (derive_medium, host_scene.cpp)
  row[2] = mk4(extinction, __uint_as_float(0u));                     // extinction, cls
  row[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                      // density (none), bounds_min.xy
  row[4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                      // bounds_min.z, bounds_max
  row[5] = make_float4(__uint_as_float(kInvalid), __uint_as_float(kInvalid), __uint_as_float(0u), __uint_as_float(0u));  // absorption_index, scattering_index, cls, explicit_connections
  row[6] = make_float4(fmaxf(extinction.x, fmaxf(extinction.y, extinction.z)), 0.0f, 0.0f, 0.0f);                        // max_sigma, dims
  row[7] = make_float4(__uint_as_float(kMediumStoredCoefficients), __uint_as_float(kInvalid), 0.0f, 0.0f);               // derived_color = "stored", derived_distances
  return index;
}

ETX_DEV bool bdpt_enter_subsurface(const DScene& scene, const etx_abi_material& mat, const Isect& isect, BsdfSample& bs, Sampler& smp, float wavelength, uint32_t& vertex_medium,
  uint32_t& walk_medium, uint32_t& path_medium) {
  if ((mat.subsurface.cls == 0u) || ((bs.properties & kSampleReflection) == 0u) || ((bs.properties & kSampleDiffuse) == 0u))
    return false;
  const f3 w_o = (mat.subsurface.path == 0u) ? sample_cosine_distribution(smp.next_2d(), -isect.nrm, 1.0f) : isect.w_i;
  walk_medium = scene.material_sss_medium[isect.material];
  path_medium = (walk_medium == mat.int_medium) ? walk_medium : kInvalid;
  vertex_medium = (walk_medium == mat.int_medium) ? walk_medium : scene.material_sss_medium[scene.subsurface_scatter_material];
  if (walk_medium == kSssMediumDynamic)
    walk_medium = bdpt_derive_walk_medium(scene, mat, isect, wavelength, vertex_medium);
  bs.w_o = w_o;
  bs.weight = mk3(1.0f);
  bs.pdf = fabsf(dot(w_o, isect.nrm)) / kPi;
  bs.eta = 1.0f;
  bs.medium_index = path_medium;
  bs.properties = kSampleTransmission | kSampleDiffuse | kSampleMediumChanged;
  return true;
}

// Wave-level slot reservation of the walk kernels: a wavefront takes kWalkChunk pool slots with ONE atomic and hands them to its
// lanes step by step (the atomics on a counter are serialised by its L2 channel at ~11 ns each, pipeline.h: one per lane and event
// - 50 M per iteration in a scene of subsurface objects - would take longer than the walks). What is left of the last chunk when the
// wavefront ends is marked unused (flags 0: no consumer of the pool connects to such a record).
constexpr uint32_t kWalkChunk = 256u;
struct WaveChunk {
  uint32_t next, end;  // wave-uniform
};
ETX_DEV uint32_t wave_chunk_slot(bool wanted, WaveChunk& chunk, uint32_t* counter) {
  const unsigned long long mask = __ballot(wanted);
  const uint32_t need = uint32_t(__popcll(mask));
  if (need == 0u)
    return 0u;
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32u), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
  const uint32_t left = chunk.end - chunk.next;
  uint32_t fresh = 0u;
  if (need > left) {  // wave-uniform
    if ((threadIdx.x & 63u) == 0u)
      fresh = atomicAdd(counter, kWalkChunk);
    fresh = __builtin_amdgcn_readfirstlane(fresh);
  }
  // the first `left` lanes finish the old chunk, the others start the new one
  const uint32_t slot = (rank < left) ? (chunk.next + rank) : (fresh + (rank - left));
  if (need > left)
    chunk.next = fresh + (need - left), chunk.end = fresh + kWalkChunk;
  else
    chunk.next += need;
  return slot;
}

// The three kinds of (sub-)steps. A walk is split by event class: its scattering events (free flight, material-filtered query,
// medium vertex - no BSDF, few registers, thousands of them) run in the persistent kernel k_bdpt_walk; the surface vertex where it
// leaves the object (BSDF sample of the scatter material, roulette - the out-of-line BSDF calls cost the kernel its occupancy) is
// queued and shaded densely by k_bdpt_walk_exit.
enum : uint32_t {
  kStepSegment = 0,    // a segment of the ray queue: the hit comes from the hit queue
  kStepWalkEvent = 1,  // sub-step of a walk: free flight + query; a scattering event is handled, a surface hit is handed back (r.exit, hit in `h`)
  kStepWalkExit = 2,   // the surface vertex that ends a walk
};

// What one (sub-)step of an emitter path leaves behind: the vertices to store and where the path goes.
struct BdptLightStep {
  bool store_emitter, store_vertex;
  bool alive;   // the path continues on the ray queue
  bool walking; // kStepWalkEvent: the walk continues with another sub-step; kStepSegment: the path has just entered an object (walk = which)
  bool exit;    // kStepWalkEvent: the flight reached the object's surface
  f3 v_pos, v_nrm, v_wi, v_throughput;
  float v_from_prev, v_bc_u, v_bc_v;
  uint32_t v_flags, v_tri, v_medium;
  BVtx emitter_vertex;
  float emitter_from_next;
};

// One segment of an emitter path after the closest-hit query (kInWalk = false, `h` from the hit queue), or one sub-step of a
// subsurface walk (kInWalk = true: the free flight and its material-filtered query happen here).
template <uint32_t kStep, bool kSimple, class Nodes>
ETX_DEV BdptLightStep bdpt_light_step(const DScene& scene, const Nodes& nodes, const LaneStack& stack, uint32_t mode, BdptState& st, BdptWalk& walk, float4& h) {
  constexpr bool kInWalk = kStep != kStepSegment;
  BdptLightStep r = {};
  MediumSample ms;
  ms.sampled_medium_t = 0.0f;
  bool flight_ok = true;
  if (kStep == kStepWalkEvent) {
    walk.events += 1u;
    flight_ok = (walk.events <= 1024u) && bdpt_walk_flight(scene, nodes, stack, walk, st, h, ms);  // :776-812
  }
  const uint32_t tri = __float_as_uint(h.w);
  const bool found = tri != kInvalid;
  // regular_step, bidirectional.cxx:712-727
  if ((kInWalk == false) && (st.medium != kInvalid)) {
    ms = sample_medium_homogeneous(scene, scene.mediums[st.medium], st.wavelength, st.throughput, st.sampler, st.ray_o, st.ray_d, found ? h.z : kMaxFloat);
    st.throughput *= ms.weight;
  }
  const bool first = (st.flags & kBpFirst) != 0u;
  if (flight_ok == false) {
    // the walk ended the path
  } else if (ms.sampled_medium()) {  // handle_medium, :533-570
    const uint32_t medium_index = kInWalk ? walk.medium : st.medium;
    const DMedium& med = scene.mediums[medium_index];
    const bool explicit_connections = (kInWalk == false) && (med.explicit_connections != 0u);  // subsurface_step passes false (:811)
    const f2 rnd_bsdf = st.sampler.next_2d();
    (void)st.sampler.next_2d();
    (void)st.sampler.next_2d();
    const f3 w_o = sample_phase_function(st.ray_d, med.g, rnd_bsdf);
    const float pdf_fwd = phase_function(st.ray_d, w_o, med.g);
    const float pdf_bck = phase_function(w_o, st.ray_d, med.g);
    st.path_size += 1u;
    BVtx curr = {ms.pos, mk3(0.0f), 0.0f, 0.0f, kBvMedium | kBvConnectible | ((st.prev.flags & kBvConnectible) ? kBvMisConnectible : 0u), kInvalid};
    curr.from_prev = bdpt_to_area(st.pdf_dir, st.prev.pos, curr);
    float prev_from_next = bdpt_to_area(pdf_bck, ms.pos, st.prev);
    if (first && (st.flags & kBpDistantEmitter)) {  // update_distant_emitter_path_pdfs, :423-436
      st.prev.from_prev = bdpt_emitter_sample_pdf(scene, scene.emitters[st.flags >> kBpEmitterShift], -st.ray_d);
      curr.from_prev = st.aux;
    }
    r.v_pos = ms.pos, r.v_wi = st.ray_d, r.v_throughput = st.throughput, r.v_from_prev = curr.from_prev, r.v_flags = curr.flags, r.v_medium = medium_index, r.v_tri = kInvalid;
    if (explicit_connections == false)
      r.v_flags |= kBvNoCameraConnection;  // handle_medium skips connect() for it; camera paths still connect TO it
    st.ray_o = ms.pos, st.ray_d = w_o, st.ray_tmin = kRayEpsilon, st.ray_tmax = kMaxFloat;
    st.pdf_dir = pdf_fwd;
    r.emitter_vertex = st.prev;
    r.emitter_from_next = prev_from_next;
    bdpt_advance_history(st, prev_from_next, false, mode, true);
    r.emitter_vertex.history = st.prev.history;
    r.store_emitter = first, r.store_vertex = true;
    st.prev = curr;
    st.flags &= ~kBpFirst;
    if (kInWalk)
      r.walking = true;  // no roulette inside the walk (:776-815)
    else
      r.alive = random_continue(st.path_size - 2u, scene.random_path_termination, st.eta, st.sampler, st.throughput) && (st.path_size - 1u < scene.max_path_length);
  } else if (found) {
   if constexpr (kStep == kStepWalkEvent) {
    r.exit = true;
   } else {
    Isect isect = make_intersection(scene, st.ray_d, h.x, h.y, h.z, tri);
    if (kInWalk)
      isect.material = scene.subsurface_scatter_material;  // build_path :858-861
    const etx_abi_material& mat = scene.materials[isect.material];
    const f2 rnd_bsdf = st.sampler.next_2d();
    (void)st.sampler.next_2d();
    const f2 rnd_support = st.sampler.next_2d();
    if (mat.cls == ETX_MAT_BOUNDARY) {  // handle_surface :586-593: the path length does not change
      const etx_abi_triangle& t = scene.triangles[isect.tri];
      st.medium = (dot(isect.geo_n, st.ray_d) < 0.0f) ? mat.int_medium : mat.ext_medium;
      st.ray_o = shading_pos(scene, t, isect.bc, st.ray_d);
      st.ray_tmin = kRayEpsilon, st.ray_tmax = kMaxFloat;
      r.alive = true;
    } else {
      const BsdfData data = make_bsdf_data(isect, isect.w_i, st.medium, kPathLight, st.wavelength);
      st.sampler.push_fixed(rnd_bsdf.x, rnd_bsdf.y, rnd_support.x);
      BsdfSample bs = bsdf_sample_s<kSimple>(scene, data, mat, st.sampler);
      st.sampler.pop_fixed();
      uint32_t vertex_medium = (bs.properties & kSampleMediumChanged) ? bs.medium_index : st.medium;
      uint32_t walk_medium = kInvalid, path_medium = vertex_medium;
      const bool enter = (kInWalk == false) && bdpt_enter_subsurface(scene, mat, isect, bs, st.sampler, st.wavelength, vertex_medium, walk_medium, path_medium);
      const uint32_t vertex_material = enter ? scene.subsurface_scatter_material : isect.material;
      st.path_size += 1u;
      const bool connectible = (bs.properties & kSampleDelta) == 0u;
      BVtx curr = {isect.pos, isect.nrm, 0.0f, 0.0f,
        kBvSurface | (connectible ? kBvConnectible : 0u) | ((connectible && (st.prev.flags & kBvConnectible)) ? kBvMisConnectible : 0u), isect.tri};
      curr.from_prev = bdpt_to_area(st.pdf_dir, st.prev.pos, curr);
      const float rev_pdf = bdpt_reverse_pdf<kSimple>(scene, data, bs.w_o, scene.materials[vertex_material], st.sampler);
      const float prev_from_next = bdpt_to_area(rev_pdf, isect.pos, st.prev);
      r.v_pos = isect.pos, r.v_nrm = isect.nrm, r.v_wi = isect.w_i, r.v_throughput = st.throughput, r.v_flags = curr.flags, r.v_tri = isect.tri, r.v_bc_u = isect.bc.y, r.v_bc_v = isect.bc.z;
      if (enter || kInWalk)
        r.v_flags |= kBvScatterMaterial;
      if (scene.material_general_bsdf[vertex_material] != 0u)
        r.v_flags |= kBvGeneralBsdf;
      r.v_medium = vertex_medium;
      st.medium = path_medium;
      bool terminate = false;
      if (bs.valid()) {
        st.pdf_dir = bs.pdf;
        st.throughput *= bs.weight;
        const etx_abi_triangle& t = scene.triangles[isect.tri];
        st.ray_o = shading_pos(scene, t, isect.bc, bs.w_o);
        st.ray_d = bs.w_o;
        st.ray_tmin = kRayEpsilon, st.ray_tmax = kMaxFloat;
        st.throughput *= fix_shading_normal(isect.geo_n, isect.nrm, isect.w_i, bs.w_o);
      } else {
        terminate = true;
      }
      if (first && (st.flags & kBpDistantEmitter)) {  // update_distant_emitter_path_pdfs
        st.prev.from_prev = bdpt_emitter_sample_pdf(scene, scene.emitters[st.flags >> kBpEmitterShift], -isect.w_i);
        curr.from_prev = st.aux * fabsf(dot(isect.w_i, isect.geo_n));
      }
      r.v_from_prev = curr.from_prev;
      r.emitter_vertex = st.prev;
      r.emitter_from_next = prev_from_next;
      bdpt_advance_history(st, prev_from_next, false, mode, connectible);
      r.emitter_vertex.history = st.prev.history;
      r.store_emitter = first, r.store_vertex = true;
      st.prev = curr;
      st.flags &= ~kBpFirst;
      r.alive = (terminate == false) && random_continue(st.path_size - 2u, scene.random_path_termination, st.eta, st.sampler, st.throughput) && (st.path_size - 1u < scene.max_path_length);
      // a surface vertex ends a walk (:858-861) or starts one: the path goes to the walk queue and joins the ray queue again when it has left the object
      if (enter && r.alive) {
        walk = {isect.material, walk_medium, 0u};
        r.alive = false;
        r.walking = true;
      }
    }
   }
  }
  return r;
}

// the pool records of a (sub-)step, slots reserved by the caller
ETX_DEV void bdpt_light_store(const Pipeline& p, BdptState& st, const BdptLightStep& r, uint32_t emitter_slot, uint32_t vertex_slot) {
  if (r.store_emitter)
    bdpt_store_light_vertex(p, emitter_slot, st.id, r.emitter_vertex.pos, r.emitter_vertex.nrm, mk3(0.0f), mk3(0.0f), r.emitter_vertex.from_prev, r.emitter_vertex.history, r.emitter_vertex.flags,
      r.emitter_vertex.tri, r.emitter_from_next, 0.0f, 0u, 1u, kInvalid, kInvalid, st.wavelength, 0u);
  if (r.store_vertex) {
    const uint32_t previous = r.store_emitter ? emitter_slot : st.prev_slot;
    Sampler derived;
    derived.init(st.sampler.seed, 0x62647074u);
    // history of the new vertex = the path's running history now (dev_bdpt.h)
    bdpt_store_light_vertex(p, vertex_slot, st.id, r.v_pos, r.v_nrm, r.v_wi, r.v_throughput, r.v_from_prev, st.mis_history, r.v_flags, r.v_tri, r.v_bc_u, r.v_bc_v, st.path_size - 1u, st.path_size, r.v_medium,
      previous, st.wavelength, derived.seed);
    st.prev_slot = vertex_slot;
  }
}

ETX_DEV void bdpt_walk_push(const Pipeline& p, uint32_t queue, uint32_t slot, const BdptState& st, uint32_t prev_w, const BdptWalk& walk) {
  if (slot >= p.capacity)
    return;  // cannot happen: every path is in one place (path set, walk queue or exit queue)
  bdpt_store(p.walk[queue], slot, st, prev_w);
  p.walk_info[queue][slot] = make_uint2(walk.material | (walk.events << 20u), walk.medium);  // material < 2^20 (host_scene.cpp), events <= 1024; the medium index needs all 32 bits (per-walk rows)
}
ETX_DEV BdptWalk bdpt_walk_info(const Pipeline& p, uint32_t queue, uint32_t entry) {
  const uint2 info = p.walk_info[queue][entry];
  return {info.x & 0xfffffu, info.y, info.x >> 20u};
}

// Mixed scenes (DeviceScene::bdpt_binning: most surfaces Lambert, some of a class that needs the out-of-line BSDF library - the plastic coat of a subsurface
// object next to diffuse walls): until round 6 every kernel that evaluates a BSDF ran its general instantiation for EVERY item of such a scene (256 VGPRs + 110 AGPRs,
// 784 B of call frames per lane, one wavefront per SIMD). Now a kernel exists in three parts: kPartAll (scenes of one kind: every item, the instantiation the
// scene asks for), and for mixed scenes kPartSimple - the inline-Lambert instantiation over every item, which hands the items whose material needs the general
// library (DScene::material_general_bsdf) to a list - followed by kPartGeneral, the general instantiation over that list, dense. As the VCM / PT shade kernels
// bin by shading group (kernels_shade.inl); the lists and their counters are theirs (free under this integrator; cleared per round by round_housekeeping).
enum : uint32_t { kPartAll = 0u, kPartSimple = 1u, kPartGeneral = 2u };

ETX_DEV bool bdpt_hit_general(const DScene& scene, const float4& h) {
  const uint32_t tri = __float_as_uint(h.w);
  return (tri != kInvalid) && (scene.material_general_bsdf[scene.triangles[tri].material_index] != 0u);
}

// kPartSimple: true = the item stays with this kernel. Workgroup-uniform call (block compaction); `list` holds one entry per live path at most.
ETX_DEV bool bdpt_bin_general(const Pipeline& p, uint32_t list, uint32_t item, bool valid, bool general, BlockScratch& scratch) {
  const bool hand_over = valid && general;
  const uint32_t at = block_compact_slot(hand_over, p.counters + (list == 0u ? kCntGroupGeneral : kCntGroupSubsurface), scratch);
  if (hand_over) {
    if (at < p.capacity)
      p.group_list[list][at] = item;
    else
      atomicOr(p.counters + kCntOverflow, kOverflowLightVertices);  // cannot happen (see the callers); if it did, the iteration would be discarded, not rendered short
  }
  return valid && (general == false);
}

// One segment of an emitter path after the closest-hit query
template <bool kSimple, uint32_t kPart>
__global__ __launch_bounds__(kBlockSize) void k_bdpt_light_shade(Pipeline p, VcmParams it, uint32_t in_set) {
  static_assert((kPart == kPartAll) || (kSimple == (kPart == kPartSimple)), "kPartSimple runs the inline instantiation, kPartGeneral the out-of-line one");
  __shared__ BlockScratch s_scratch;
  const LaneStack no_stack = {};
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const PathSet& out = p.paths[in_set ^ 1u];
  const uint32_t count = (kPart == kPartGeneral) ? min(p.counters[kCntGroupGeneral], p.capacity) : p.counters[in_set == 0 ? kCntActiveA : kCntActiveB];
  uint32_t* out_counter = p.counters + (in_set == 0 ? kCntActiveB : kCntActiveA);
  const uint32_t mode = bdpt_mode(it);
  ETX_BLOCK_LOOP(count, j) {
    bool valid = j < count;
    const uint32_t i = (kPart == kPartGeneral) ? (valid ? p.group_list[0][j] : 0u) : j;
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (valid)
      h = p.hits[i];
    if (kPart == kPartSimple)
      valid = bdpt_bin_general(p, 0u, i, valid, bdpt_hit_general(scene, h), s_scratch);
    BdptState st = {};
    BdptWalk walk = {kInvalid, kInvalid, 0u};
    BdptLightStep r = {};
    if (valid) {
      st = bdpt_load(in, i);
      if (st.flags & kBpFirst) {
        st.prev.tri = st.prev_slot;  // see k_bdpt_light_generate
        st.prev_slot = kInvalid;
      }
      r = bdpt_light_step<kStepSegment, kSimple>(scene, global_nodes(scene), no_stack, mode, st, walk, h);
    }
    // pool slots: the emitter vertex (first interaction only), then the new vertex
    const uint32_t emitter_slot = block_compact_slot(r.store_emitter, p.counters + kCntLightVertices, s_scratch);
    const uint32_t vertex_slot = block_compact_slot(r.store_vertex, p.counters + kCntLightVertices, s_scratch);
    bdpt_light_store(p, st, r, emitter_slot, vertex_slot);
    const uint32_t slot = block_compact_slot(r.alive, out_counter, s_scratch);
    if (r.alive)
      bdpt_store(out, slot, st, (st.flags & kBpFirst) ? st.prev.tri : st.prev_slot);
    if (p.walk_info[0] != nullptr) {  // kernel-uniform: the scene has subsurface materials
      const uint32_t walk_slot = block_compact_slot(r.walking, p.counters + kCntWalk + 32u * in_set, s_scratch);
      if (r.walking)
        bdpt_walk_push(p, in_set, walk_slot, st, st.prev_slot, walk);
    }
  }
}

// The subsurface walks of the paths that entered an object in this bounce (walk queue -> "out" path set). Persistent wavefronts:
// a lane runs the sub-steps of ITS walk; whenever kWalkRefill lanes of a wavefront are idle they take the next entries of the queue.
#if !defined(ETX_WALK_REFILL)  // (experiment builds: tools/build_variant.sh ... "-DETX_WALK_REFILL=8 -DETX_WALK_BUDGET=64" kernels_bdpt.hip)
#define ETX_WALK_REFILL 16u
#endif
#if !defined(ETX_WALK_BUDGET)
#define ETX_WALK_BUDGET 32u
#endif
constexpr uint32_t kWalkRefill = ETX_WALK_REFILL;
// (256, one per CU: each workgroup holds 40 KB of LDS for its whole, persistent life; four per CU filled every CU's LDS and kept the other lanes' traversal kernels out -
// configs[3], six lanes: 2048 workgroups 38.5, 1024 38.8, 512 39.4, 256 40.2 Msamples/s, profiles/round6_ab_walk_blocks.txt)
#if !defined(ETX_WALK_BLOCKS)
#define ETX_WALK_BLOCKS 256u
#endif
constexpr uint32_t kWalkBlocks = ETX_WALK_BLOCKS;
// Scattering events a walk gets per round. Most walks leave their object after a few events, a few take hundreds (the reference
// allows 1024, :776): a kernel that ran every walk of a bounce to its end lasted as long as its longest walk (measured 7 ms per
// launch, 100 ms per iteration, with nearly all lanes idle). A walk that is still inside after its budget goes to the other walk
// queue and continues in the next round, next to that round's new walks.
constexpr uint32_t kWalkBudget = ETX_WALK_BUDGET;
// (Tried: letting the last <= 4096 walks of a pass run to their end in one launch - fewer rounds, 111 -> 91 per iteration, but the
// launch then lasts as long as its longest walk with the stream idle behind it: 23.9 vs 25.3 Msamples/s. Not adopted.)
constexpr uint32_t kWalkLdsNodes = 64u;  // 8 KB next to the 32 KB of stacks: four workgroups per CU

// Taking the next entries of the walk queue: called by all lanes of a wavefront; lanes without a walk get one while the queue lasts.
// Returns false when the wavefront has nothing left to do.
ETX_DEV bool walk_refill(const Pipeline& p, uint32_t count, bool& active, bool& exhausted, uint32_t& entry) {
  const unsigned long long idle_mask = __ballot(active == false);
  const uint32_t idle_count = uint32_t(__popcll(idle_mask));
  entry = kInvalid;
  if ((exhausted == false) && ((idle_count >= kWalkRefill) || (idle_count == 64u))) {
    uint32_t base = 0u;
    if ((threadIdx.x & 63u) == 0u)
      base = atomicAdd(p.counters + kCntWalkFetch, idle_count);
    base = __builtin_amdgcn_readfirstlane(base);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(idle_mask >> 32u), __builtin_amdgcn_mbcnt_lo(uint32_t(idle_mask), 0u));
    if ((active == false) && (base + rank < count))
      entry = base + rank;
    exhausted = base + idle_count >= count;
  }
  return (__ballot(active || (entry != kInvalid)) != 0ull) || (exhausted == false);
}

// The scattering events of the walks of this bounce, emitter paths: walk queue -> (medium vertices in the light vertex pool) -> exit queue
__global__ __launch_bounds__(kBlockSize, 4) void k_bdpt_walk_light(Pipeline p, VcmParams it, uint32_t queue) {
  __shared__ int32_t s_stack[kStackDepth * kBlockSize];
  __shared__ float4 s_nodes[kWalkLdsNodes * 8u];
  const LaneStack stack = lane_stack(p.scene, s_stack + threadIdx.x, kBlockSize);
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntWalk + 32u * queue], p.capacity);
  if (count == 0u)
    return;
  const BvhNodes nodes = stage_nodes(scene, s_nodes, kWalkLdsNodes);  // every event descends from the root: its first levels come from LDS
  const uint32_t mode = bdpt_mode(it);
  const uint32_t lane = threadIdx.x & 63u;
  BdptState st = {};
  BdptWalk walk = {kInvalid, kInvalid, 0u};
  WaveChunk chunk = {0u, 0u};
  bool active = false, exhausted = false;
  uint32_t entry = kInvalid, budget = 0u;
  while (walk_refill(p, count, active, exhausted, entry)) {
    if (entry != kInvalid) {
      st = bdpt_load(p.walk[queue], entry);
      walk = bdpt_walk_info(p, queue, entry);
      budget = kWalkBudget;
      active = true;
    }
    BdptLightStep r = {};
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (active)
      r = bdpt_light_step<kStepWalkEvent, true>(scene, nodes, stack, mode, st, walk, h);
    const uint32_t vertex_slot = wave_chunk_slot(r.store_vertex, chunk, p.counters + kCntLightVertices);
    bdpt_light_store(p, st, r, 0u, vertex_slot);  // a walk never holds the emitter vertex: its entry vertex came first
    const uint32_t exit_slot = wave_compact_slot(r.exit, p.counters + kCntWalkExit);
    if (r.exit && (exit_slot < p.capacity)) {
      bdpt_store(p.walk_exit, exit_slot, st, st.prev_slot);
      p.walk_exit_hits[exit_slot] = h;
    }
    active = active && r.walking;
    // out of budget: the walk continues in the next round
    budget -= active ? 1u : 0u;
    const bool later = active && (budget == 0u);
    const uint32_t later_slot = wave_compact_slot(later, p.counters + kCntWalk + 32u * (queue ^ 1u));
    if (later)
      bdpt_walk_push(p, queue ^ 1u, later_slot, st, st.prev_slot, walk);
    active = active && (later == false);
  }
  // the unused rest of this wavefront's last chunk: records nobody may connect to
  for (uint32_t i = chunk.next + lane; i < min(chunk.end, p.lv.capacity); i += 64u) {
    p.lv.thr_dvm(i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    p.lv.bc_len_med(i) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
  }
}

// ... and the surface vertices where they leave their objects: exit queue -> "out" path set
template <bool kSimple>
__global__ __launch_bounds__(kBlockSize) void k_bdpt_walk_exit_light(Pipeline p, VcmParams it, uint32_t out_set) {
  __shared__ BlockScratch s_scratch;
  const LaneStack no_stack = {};
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntWalkExit], p.capacity);
  const PathSet& out = p.paths[out_set];
  uint32_t* out_counter = p.counters + (out_set == 0 ? kCntActiveA : kCntActiveB);
  const uint32_t mode = bdpt_mode(it);
  ETX_BLOCK_LOOP(count, i) {
    BdptState st = {};
    BdptWalk walk = {kInvalid, kInvalid, 0u};
    BdptLightStep r = {};
    if (i < count) {
      st = bdpt_load(p.walk_exit, i);
      float4 h = p.walk_exit_hits[i];
      r = bdpt_light_step<kStepWalkExit, kSimple>(scene, global_nodes(scene), no_stack, mode, st, walk, h);
    }
    const uint32_t vertex_slot = block_compact_slot(r.store_vertex, p.counters + kCntLightVertices, s_scratch);
    bdpt_light_store(p, st, r, 0u, vertex_slot);
    const uint32_t slot = block_compact_slot(r.alive, out_counter, s_scratch);
    if (r.alive)
      bdpt_store(out, slot, st, st.prev_slot);
  }
}

// connect_light_to_camera (:1380-1428) for the vertices the light pass stored in this bounce
// (kPartSimple hands the vertices flagged kBvGeneralBsdf to list 1: surface vertices only, one per path and round - the entry and the exit vertex of a walk carry
// the scatter material, which is of a simple class whenever the host enables the binning)
template <bool kSimple, uint32_t kPart>
__global__ __launch_bounds__(kBlockSize) void k_bdpt_connect_camera(Pipeline p, VcmParams it) {
  static_assert((kPart == kPartAll) || (kSimple == (kPart == kPartSimple)), "part / instantiation");
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  const uint32_t begin = p.counters[kCntLightBounceBegin];
  const uint32_t end = min(p.counters[kCntLightVertices], p.lv.capacity);
  const uint32_t count = (kPart == kPartGeneral) ? min(p.counters[kCntGroupSubsurface], p.capacity) : ((end > begin) ? (end - begin) : 0u);
  const uint32_t mode = bdpt_mode(it);
  ETX_BLOCK_LOOP(count, j) {
    ShadowRequest request;
    bool queue = false;
    bool valid = j < count;
    const uint32_t vi = (kPart == kPartGeneral) ? (valid ? p.group_list[1][j] : 0u) : (begin + j);
    uint32_t flags = 0u;
    if (valid) {
      flags = __float_as_uint(p.lv.thr_dvm(vi).w);
      const uint32_t index_in_path = __float_as_uint(p.lv.bc_len_med(vi).z) >> 16u;
      valid = (flags & kBvConnectible) && ((flags & kBvNoCameraConnection) == 0u) && (index_in_path >= 1u) && opt_connect_to_camera(it);
    }
    if (kPart == kPartSimple)
      valid = bdpt_bin_general(p, 1u, vi, valid, (flags & kBvGeneralBsdf) != 0u, s_scratch);
    if (valid) {
      {
        BdptLightVertex y = bdpt_load_light_vertex(p, scene, vi);
        const uint32_t target_path_length = y.path_size;  // emitter_path_length() + 1
        if ((target_path_length <= scene.max_path_length) && (target_path_length >= scene.min_path_length)) {
          Sampler smp;
          smp.seed = y.seed, smp.fixed_u = smp.fixed_v = smp.fixed_w = 0.0f;
          const CameraSample cs = sample_film(scene, smp, y.self.pos);
          const float len = length(cs.position - y.self.pos);
          const float cos_t = fabsf(dot(cs.direction, scene.camera.direction));
          const float near_extent = (scene.camera.clip_near > 0.0f) ? scene.camera.clip_near / cos_t : 0.0f;
          const float far_extent = (scene.camera.clip_far > 0.0f) ? scene.camera.clip_far / cos_t : kMaxFloat;
          if ((cs.pdf_dir > 0.0f) && (len >= near_extent) && (len <= far_extent)) {
            const BdptBsdf bsdf = bdpt_bsdf<kSimple>(scene, y.full, kPathLight, cs.direction, y.wavelength, smp);
            if (is_zero(bsdf.bsdf) == false) {
              float weight = 1.0f;
              if (opt_enable_mis(it) && (mode != kBdptLightTracing)) {  // mis_weight_light_to_camera, :1135-1182 (Full)
                const BVtx y_prev = bdpt_load_light_summary(p, y.prev);
                const BVtx camera_vertex = {cs.position, cs.normal, 0.0f, 0.0f, 0u, kInvalid};
                const float cos_c = dot(normalize(y.self.pos - scene.camera.position), scene.camera.direction);
                const float film_pdf = 1.0f / fabsf(scene.camera.area * cos_c * cos_c * cos_c);  // film_pdf_out, scene_camera.hxx:20-24
                const float curr_from_camera = bdpt_to_area(film_pdf, cs.position, y.self);
                const float prev_from_curr = bdpt_pdf_area<kSimple>(scene, kPathCamera, cs.position, y.full, y_prev, y.wavelength, smp);
                if (mode == kBdptFast) {
                  const uint32_t* table = reinterpret_cast<const uint32_t*>(p.light_path_table) + size_t(y.path) * p.path_table_entries + kBdptRowHeader;
                  const uint32_t e0 = table[0] & ~kPathEntryGeneralBit, e1 = table[1] & ~kPathEntryGeneralBit;
                  const float p_sample = p.lv.pos_dvcm(e0).w;  // e0.pdf.from_prev
                  const uint32_t e0_flags = __float_as_uint(p.lv.thr_dvm(e0).w), e1_flags = __float_as_uint(p.lv.thr_dvm(e1).w);
                  float p_light = y_prev.from_prev * y.self.from_prev;
                  float p_bck = curr_from_camera * y_prev.history;
                  float p_direct = prev_from_curr;
                  if (y.path_size > 2u) {
                    p_direct = p.lv.bc_len_med(e0).x;  // e0.pdf.from_next: the connection of the first vertex to the emitter
                    p_bck *= prev_from_curr;
                    p_light *= p_sample;
                  }
                  const float p_camera_direct = (e0_flags & kBvMisConnectible) ? p_bck * p_direct : 0.0f;
                  const float p_camera_connect = (e1_flags & kBvConnectible) ? p_bck * p_sample : 0.0f;
                  weight = balance_heuristic(p_light, p_camera_direct, p_camera_connect);
                } else {
                  weight = 1.0f / (1.0f + bdpt_mis_light(curr_from_camera, y.self.from_prev, prev_from_curr, y_prev));
                }
              }
              const f3 splat = y.throughput * bsdf.bsdf * (cs.weight * weight) * spectral_film_weight(scene, y.wavelength);
              const uint32_t x = static_cast<uint32_t>((cs.uv.x * 0.5f + 0.5f) * float(it.film_w));
              const uint32_t yy = static_cast<uint32_t>((cs.uv.y * 0.5f + 0.5f) * float(it.film_h));
              if ((x < it.film_w) && (yy < it.film_h) && (is_zero(splat) == false)) {
                const f3 clip_pos = y.self.pos + cs.direction * fmaxf(0.0f, len - near_extent);
                request = {bdpt_segment_origin(scene, y.full, clip_pos), clip_pos, splat, y.medium, kShadowTargetLight | (x + (it.film_h - 1u - yy) * it.film_w), y.wavelength};
                queue = true;
              }
            }
          }
        }
      }
    }
    const uint32_t slot = block_compact_slot(queue, p.counters + kCntShadow, s_scratch);
    if (queue)
      write_shadow(p, slot, request);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// build_camera_path, bidirectional.cxx:900-953 (+ execute_range :366-409)
__global__ __launch_bounds__(kBlockSize) void k_bdpt_camera_generate(Pipeline p, VcmParams it) {
  const DScene& scene = p.scene;
  const uint32_t mode = bdpt_mode(it);
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < it.path_count; k += gridDim.x * blockDim.x) {
    const uint32_t i = path_pixel(it, k);
    BdptState st = {};
    st.id = i;
    // the reference seeds the camera sampler like the light sampler of the same pixel (:377-378); the device gives the
    // camera path a stream of its own (as for VCM, kernels_vcm.hip k_camera_generate and DESIGN.md 4)
    st.sampler.init(i, it.iteration);
    if ((it.options & kOptionReferenceSeeding) == 0u)  // reference_seeding: shared streams, see k_camera_generate
      st.sampler.seed = Sampler::random_seed(st.sampler.seed, 0x43414d45u);
    st.wavelength = 0.0f;
    if (scene.spectral) {
      const float u = st.sampler.next();
      st.wavelength = (mode == kBdptPathTracing) ? spectral_sample_wavelength(u) : p.path_wavelength[i];
    }
    const uint32_t px = i % it.film_w, py = i / it.film_w;
    const f2 uv = bdpt_film_sample(scene, it.iteration != 0u, px, py, it, st.sampler.next_2d());
    const RayGen r = generate_ray(scene, uv, st.sampler.next_2d());
    st.ray_o = r.o, st.ray_d = r.d, st.ray_tmin = r.tmin, st.ray_tmax = r.tmax;
    st.throughput = mk3(1.0f);
    st.eta = 1.0f;
    st.pdf_dir = film_evaluate_out_pdf_dir(scene, st.ray_d);
    st.mis_history = (mode == kBdptFast) ? 1.0f : 0.0f;  // :925-931
    st.aux = 0.0f;
    st.path_size = 1u;
    st.medium = scene.camera.medium_index;
    st.flags = kBpFirst | ((it.options & kOptionRetryKeepsAovs) ? kBpGBuffer : 0u);
    st.prev = {r.o, scene.camera.direction, 1.0f, 0.0f, kBvConnectible | kBvMisConnectible, kInvalid};
    bdpt_store(p.paths[0], k, st, kInvalid);
  }
  if ((blockIdx.x == 0) && (threadIdx.x == 0)) {
    p.counters[kCntActiveA] = it.path_count;
    p.counters[kCntWalk] = p.counters[kCntWalk + 32u] = 0u;  // the light pass has drained its walk queues; their last counts are stale
  }
}

// mis_weight_direct_hit, bidirectional.cxx:1211-1233
ETX_DEV float bdpt_direct_hit_weight(const BdptState& st, uint32_t mode, float z_curr_from_prev, float p_sample, float p_from) {
  if (mode == kBdptFast) {
    const float to_emitter_direct = st.prev.from_prev * z_curr_from_prev;
    const float to_emitter_connect = (st.prev.flags & kBvConnectible) ? st.prev.from_prev * p_sample : 0.0f;
    return balance_heuristic(to_emitter_direct, to_emitter_connect, st.prev.history * (p_from * p_sample));
  }
  return 1.0f / (1.0f + bdpt_mis_camera(st.path_size, p_sample, z_curr_from_prev, p_from, st.prev));
}

// What one (sub-)step of a camera path leaves behind.
struct BdptCameraStep {
  bool store_vertex;    // a connectible vertex for the camera vertex pool (written BEFORE `created` replaces st.prev: the record reads z_prev)
  bool created;         // a path vertex (curr) exists and becomes prev
  bool terminate, enter, scatter_vertex, in_medium_event, general_bsdf;
  bool exit;            // kStepWalkEvent: the flight reached the object's surface (hit in `h`)
  bool alive;           // boundary crossing: the path continues on the ray queue without a vertex
  BVtx curr;
  float4 v_hit;
  f3 v_wi, v_throughput, v_rnd;
  uint32_t v_medium, enter_material, enter_medium;
};

// One segment of a camera path after the closest-hit query (kInWalk = false), or one sub-step of a subsurface walk (kInWalk = true).
// Film contributions of direct hits are added here; pool records and the roulette are the caller's (bdpt_camera_finish).
template <uint32_t kStep, bool kSimple, class Nodes>
ETX_DEV BdptCameraStep bdpt_camera_step(const Pipeline& p, const DScene& scene, const Nodes& nodes, const LaneStack& stack, const VcmParams& it, uint32_t mode, bool use_mis, BdptState& st, BdptWalk& walk, float4& h) {
  constexpr bool kInWalk = kStep != kStepSegment;
  BdptCameraStep r = {};
  r.v_hit = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
  r.v_medium = r.enter_material = r.enter_medium = kInvalid;
  MediumSample ms;
  ms.sampled_medium_t = 0.0f;
  bool flight_ok = true;
  if (kStep == kStepWalkEvent) {
    walk.events += 1u;
    flight_ok = (walk.events <= 1024u) && bdpt_walk_flight(scene, nodes, stack, walk, st, h, ms);  // subsurface_step, :776-812
  }
  const uint32_t tri = __float_as_uint(h.w);
  const bool found = tri != kInvalid;
  const uint32_t film_target = film_index(it, st.id);
  const f3 film_weight = spectral_film_weight(scene, st.wavelength);
  if ((kInWalk == false) && (st.medium != kInvalid)) {
    ms = sample_medium_homogeneous(scene, scene.mediums[st.medium], st.wavelength, st.throughput, st.sampler, st.ray_o, st.ray_d, found ? h.z : kMaxFloat);
    st.throughput *= ms.weight;
  }
  const bool first = (st.flags & kBpFirst) != 0u;
  if (flight_ok == false) {
    // the walk ended the path
  } else if (ms.sampled_medium()) {  // handle_medium, :533-570
    const uint32_t medium_index = kInWalk ? walk.medium : st.medium;
    const DMedium& med = scene.mediums[medium_index];
    f2 rnd_bsdf = st.sampler.next_2d();
    f2 rnd_em = st.sampler.next_2d();
    f2 rnd_support = st.sampler.next_2d();
    if ((it.bluenoise != nullptr) && first && (it.iteration < 256u))
      bluenoise_samples(it.bluenoise, st.id % it.film_w, st.id / it.film_w, it.iteration, rnd_bsdf, rnd_em, rnd_support);
    const f3 w_o = sample_phase_function(st.ray_d, med.g, rnd_bsdf);
    const float pdf_fwd = phase_function(st.ray_d, w_o, med.g);
    const float pdf_bck = phase_function(w_o, st.ray_d, med.g);
    st.path_size += 1u;
    r.curr = {ms.pos, mk3(0.0f), 0.0f, 0.0f, kBvMedium | kBvConnectible | ((st.prev.flags & kBvConnectible) ? kBvMisConnectible : 0u), kInvalid};
    r.curr.from_prev = bdpt_to_area(st.pdf_dir, st.prev.pos, r.curr);
    const float prev_from_next = bdpt_to_area(pdf_bck, ms.pos, st.prev);
    r.v_hit = mk4(ms.pos, __uint_as_float(kInvalid)), r.v_wi = st.ray_d, r.v_throughput = st.throughput, r.v_medium = medium_index;
    r.v_rnd = {rnd_em.x, rnd_em.y, rnd_support.y};
    st.ray_o = ms.pos, st.ray_d = w_o, st.ray_tmin = kRayEpsilon, st.ray_tmax = kMaxFloat;
    st.pdf_dir = pdf_fwd;
    bdpt_advance_history(st, prev_from_next, true, mode, true);
    r.created = true;
    r.in_medium_event = true;
    r.store_vertex = (kInWalk == false) && (med.explicit_connections != 0u) && (mode != kBdptLightTracing);  // subsurface_step: no explicit connections (:811)
  } else if (found) {
   if constexpr (kStep == kStepWalkEvent) {
    r.exit = true;
   } else {
    Isect isect = make_intersection(scene, st.ray_d, h.x, h.y, h.z, tri);
    if (kInWalk)
      isect.material = scene.subsurface_scatter_material;  // build_path :858-861
    r.scatter_vertex = kInWalk;
    const etx_abi_material& mat = scene.materials[isect.material];
    f2 rnd_bsdf = st.sampler.next_2d();
    f2 rnd_em = st.sampler.next_2d();
    f2 rnd_support = st.sampler.next_2d();
    if ((it.bluenoise != nullptr) && first)
      bluenoise_samples(it.bluenoise, st.id % it.film_w, st.id / it.film_w, it.iteration, rnd_bsdf, rnd_em, rnd_support);
    if (mat.cls == ETX_MAT_BOUNDARY) {  // handle_surface :586-593: no vertex, the path length does not change
      const etx_abi_triangle& t = scene.triangles[isect.tri];
      st.medium = (dot(isect.geo_n, st.ray_d) < 0.0f) ? mat.int_medium : mat.ext_medium;
      st.ray_o = shading_pos(scene, t, isect.bc, st.ray_d);
      st.ray_tmin = kRayEpsilon, st.ray_tmax = kMaxFloat;
      r.alive = true;
    } else {
      const BsdfData data = make_bsdf_data(isect, isect.w_i, st.medium, kPathCamera, st.wavelength);
      if ((st.flags & kBpGBuffer) == 0u) {  // GBuffer, :597-601
        film_add(p, p.normal_sum + film_target, isect.nrm);
        film_add(p, p.albedo_sum + film_target, bdpt_albedo(scene, mat, isect.tex, st.wavelength) * film_weight);
        st.flags |= kBpGBuffer;
      }
      st.sampler.push_fixed(rnd_bsdf.x, rnd_bsdf.y, rnd_support.x);
      BsdfSample bs = bsdf_sample_s<kSimple>(scene, data, mat, st.sampler);
      st.sampler.pop_fixed();
      uint32_t vertex_medium = (bs.properties & kSampleMediumChanged) ? bs.medium_index : st.medium;
      uint32_t path_medium = vertex_medium;
      r.enter = (kInWalk == false) && bdpt_enter_subsurface(scene, mat, isect, bs, st.sampler, st.wavelength, vertex_medium, r.enter_medium, path_medium);
      r.scatter_vertex = r.scatter_vertex || r.enter;
      r.enter_material = isect.material;
      const uint32_t vertex_material = r.enter ? scene.subsurface_scatter_material : isect.material;
      st.path_size += 1u;
      const bool connectible = (bs.properties & kSampleDelta) == 0u;
      r.curr = {isect.pos, isect.nrm, 0.0f, 0.0f, kBvSurface | (connectible ? kBvConnectible : 0u) | ((connectible && (st.prev.flags & kBvConnectible)) ? kBvMisConnectible : 0u),
        isect.tri};
      r.curr.from_prev = bdpt_to_area(st.pdf_dir, st.prev.pos, r.curr);
      const float rev_pdf = bdpt_reverse_pdf<kSimple>(scene, data, bs.w_o, scene.materials[vertex_material], st.sampler);
      const float prev_from_next = bdpt_to_area(rev_pdf, isect.pos, st.prev);
      const f3 vertex_throughput = st.throughput;
      const float prev_sampled_pdf = st.aux;  // z_prev.pdf.bsdf_sample_next (PathTracing mode weights)
      st.medium = path_medium;
      if (bs.valid()) {
        st.eta *= bs.eta;
        st.pdf_dir = bs.pdf;
        st.throughput *= bs.weight;
        st.ray_o = shading_pos(scene, scene.triangles[isect.tri], isect.bc, bs.w_o);
        st.ray_d = bs.w_o;
        st.ray_tmin = kRayEpsilon, st.ray_tmax = kMaxFloat;
      } else {
        r.terminate = true;
      }
      bdpt_advance_history(st, prev_from_next, true, mode, connectible);
      // direct_hit_area_emitter, :1235-1287 (the segment itself was the visibility query)
      if (opt_direct_hit(it) && (isect.emitter != kInvalid) && (mode != kBdptLightTracing)) {
        const uint32_t target_path_length = st.path_size - 1u;
        if ((target_path_length <= scene.max_path_length) && (target_path_length >= scene.min_path_length)) {
          const etx_abi_emitter& em = scene.emitters[isect.emitter];
          EmitterRadianceQuery q;
          q.source_position = st.prev.pos;
          q.target_position = isect.pos;
          q.direction = mk3(0.0f);
          q.uv = isect.tex;
          q.directly_visible = (st.path_size - 1u) <= 1u;
          float pdf_area = 0.0f, pdf_dir = 0.0f, pdf_dir_out = 0.0f;
          const f3 value = emitter_get_radiance(scene, em, q, pdf_area, pdf_dir, pdf_dir_out, st.wavelength);
          if (pdf_dir != 0.0f) {
            float weight = 1.0f;
            if (use_mis && (st.path_size > 2u)) {
              if (mode == kBdptPathTracing) {
                const float p_connect = pdf_dir * emitter_discrete_pdf(scene, em);
                weight = (st.prev.flags & kBvConnectible) ? power_heuristic(prev_sampled_pdf, p_connect) : 1.0f;
              } else {
                const float p_sample = bdpt_emitter_sample_pdf(scene, em, -isect.w_i);
                const float p_from = bdpt_pdf_from_emitter(scene, isect.emitter, isect.pos, isect.nrm, st.prev);
                weight = bdpt_direct_hit_weight(st, mode, r.curr.from_prev, p_sample, p_from);
              }
            }
            const f3 gathered = value * vertex_throughput * weight;
            if ((gathered.x != 0.0f) || (gathered.y != 0.0f) || (gathered.z != 0.0f))
              film_add(p, p.camera_sum + film_target, gathered * film_weight);
          }
        }
      }
      r.created = true;
      r.store_vertex = connectible && (mode != kBdptLightTracing);
      r.v_hit = h, r.v_wi = isect.w_i, r.v_throughput = vertex_throughput, r.v_medium = vertex_medium;
      r.v_rnd = {rnd_em.x, rnd_em.y, rnd_support.y};
      r.general_bsdf = scene.material_general_bsdf[vertex_material] != 0u;
    }
   }
  } else if ((kInWalk == false) && opt_direct_hit(it) && (mode != kBdptLightTracing)) {  // miss: direct_hit_environment_emitter, :1289-1340
    const float prev_sampled_pdf = st.aux;
    st.path_size += 1u;
    bdpt_advance_history(st, 0.0f, true, mode, false);
    const uint32_t target_path_length = st.path_size - 1u;
    if ((scene.env_count > 0u) && (target_path_length <= scene.max_path_length) && (target_path_length >= scene.min_path_length)) {
      f3 accumulated = mk3(0.0f);
      for (uint32_t ie = 0; ie < scene.env_count; ++ie) {
        const etx_abi_emitter& em = scene.emitters[scene.env_emitters[ie]];
        EmitterRadianceQuery q;
        q.source_position = q.target_position = mk3(0.0f);
        q.direction = st.ray_d;
        q.uv = {0.0f, 0.0f};
        q.directly_visible = (st.path_size - 1u) <= 1u;
        float pdf_area = 0.0f, pdf_dir = 0.0f, pdf_dir_out = 0.0f;
        const f3 value = emitter_get_radiance(scene, em, q, pdf_area, pdf_dir, pdf_dir_out, st.wavelength);
        float this_weight = 1.0f;
        if ((mode == kBdptPathTracing) && (st.prev.flags & kBvConnectible) && (st.path_size - 1u > 1u))
          this_weight = power_heuristic(prev_sampled_pdf, pdf_dir * emitter_discrete_pdf(scene, em));
        accumulated += value * this_weight;
      }
      if (is_zero(accumulated) == false) {
        float weight = 1.0f;
        if (use_mis && (st.path_size - 1u > 1u) && (mode != kBdptPathTracing)) {
          // pdf_for_environment_emitter, :207-222
          float pdf_dir = 0.0f;
          for (uint32_t ie = 0; ie < scene.env_count; ++ie)
            pdf_dir += bdpt_emitter_sample_pdf(scene, scene.emitters[scene.env_emitters[ie]], st.ray_d);
          pdf_dir /= float(scene.env_count);
          const float w_dot_n = st.prev.surface() ? fabsf(dot(ld3(scene.triangles[st.prev.tri].geo_n), st.ray_d)) : 1.0f;
          const float p_from = w_dot_n * env_pdf_area(scene);
          weight = bdpt_direct_hit_weight(st, mode, st.pdf_dir, pdf_dir, p_from);
        }
        film_add(p, p.camera_sum + film_target, accumulated * st.throughput * weight * film_weight);
      }
    }
    if ((st.flags & kBpGBuffer) == 0u)
      film_add(p, p.normal_sum + film_target, f3{0.0f, 0.0f, 1.0f});  // GBuffer default normal, :335-338
  } else if ((kInWalk == false) && ((st.flags & kBpGBuffer) == 0u)) {
    film_add(p, p.normal_sum + film_target, f3{0.0f, 0.0f, 1.0f});
  }
  return r;
}

// After the (sub-)step: the camera vertex record (slot reserved by the caller), prev = curr, the roulette of this interaction
// (build_path :890-895). Returns: 0 = the path ended, 1 = it continues on the ray queue, 2 = it is inside an object (walk = which;
// kInWalk: the walk goes on, else: it has just entered).
template <uint32_t kStep>
ETX_DEV uint32_t bdpt_camera_finish(const Pipeline& p, const DScene& scene, BdptState& st, BdptWalk& walk, const BdptCameraStep& r, uint32_t vertex_slot) {
  if (r.store_vertex) {
    Sampler derived;
    derived.init(st.sampler.seed, 0x51ed270bu);
    bdpt_store_camera_vertex(p, vertex_slot, st, r.v_hit, r.v_wi, r.v_medium, r.v_throughput, r.curr.from_prev, r.v_rnd, derived.seed, r.scatter_vertex, r.general_bsdf);
  }
  constexpr bool kInWalk = kStep != kStepSegment;
  if (r.created == false)
    return r.alive ? 1u : 0u;
  st.prev = r.curr;
  st.flags &= ~kBpFirst;
  st.aux = st.pdf_dir;  // becomes z_prev.pdf.bsdf_sample_next
  if (kInWalk && r.in_medium_event)
    return 2u;  // a scattering event inside the object: no roulette, the walk goes on (subsurface_step :776-815)
  const bool goes_on = (r.terminate == false) && random_continue(st.path_size - 2u, scene.random_path_termination, st.eta, st.sampler, st.throughput) && (st.path_size - 1u < scene.max_path_length);
  if (r.enter && goes_on) {  // a surface vertex ends a walk or starts one
    walk = {r.enter_material, r.enter_medium, 0u};
    return 2u;
  }
  return goes_on ? 1u : 0u;
}

template <bool kSimple, uint32_t kPart>
__global__ __launch_bounds__(kBlockSize) void k_bdpt_camera_shade(Pipeline p, VcmParams it, uint32_t in_set) {
  static_assert((kPart == kPartAll) || (kSimple == (kPart == kPartSimple)), "part / instantiation");
  __shared__ BlockScratch s_scratch;
  const LaneStack no_stack = {};
  const DScene& scene = p.scene;
  const PathSet& in = p.paths[in_set];
  const PathSet& out = p.paths[in_set ^ 1u];
  const uint32_t count = (kPart == kPartGeneral) ? min(p.counters[kCntGroupGeneral], p.capacity) : p.counters[in_set == 0 ? kCntActiveA : kCntActiveB];
  uint32_t* out_counter = p.counters + (in_set == 0 ? kCntActiveB : kCntActiveA);
  const uint32_t mode = bdpt_mode(it);
  const bool use_mis = opt_enable_mis(it);
  ETX_BLOCK_LOOP(count, j) {
    bool valid = j < count;
    const uint32_t i = (kPart == kPartGeneral) ? (valid ? p.group_list[0][j] : 0u) : j;
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    if (valid)
      h = p.hits[i];
    if (kPart == kPartSimple)
      valid = bdpt_bin_general(p, 0u, i, valid, bdpt_hit_general(scene, h), s_scratch);
    BdptState st = {};
    BdptWalk walk = {kInvalid, kInvalid, 0u};
    BdptCameraStep r = {};
    if (valid) {
      st = bdpt_load(in, i);
      st.prev.tri = st.prev_slot;  // camera paths carry the previous vertex' triangle there
      r = bdpt_camera_step<kStepSegment, kSimple>(p, scene, global_nodes(scene), no_stack, it, mode, use_mis, st, walk, h);
    }
    const uint32_t vertex_slot = block_compact_slot(r.store_vertex, p.counters + kCntCameraVertices, s_scratch);
    const uint32_t next = valid ? bdpt_camera_finish<kStepSegment>(p, scene, st, walk, r, vertex_slot) : 0u;
    const uint32_t slot = block_compact_slot(next == 1u, out_counter, s_scratch);
    if (next == 1u)
      bdpt_store(out, slot, st, st.prev.tri);
    if (p.walk_info[0] != nullptr) {  // kernel-uniform: the scene has subsurface materials
      const uint32_t walk_slot = block_compact_slot(next == 2u, p.counters + kCntWalk + 32u * in_set, s_scratch);
      if (next == 2u)
        bdpt_walk_push(p, in_set, walk_slot, st, st.prev.tri, walk);
    }
  }
}

// The scattering events of the walks of this bounce, camera paths: walk queue -> exit queue (only the exit vertex of a walk is
// connectible, :811: the events leave nothing but the path's running MIS history behind)
__global__ __launch_bounds__(kBlockSize, 4) void k_bdpt_walk_camera(Pipeline p, VcmParams it, uint32_t queue) {
  __shared__ int32_t s_stack[kStackDepth * kBlockSize];
  __shared__ float4 s_nodes[kWalkLdsNodes * 8u];
  const LaneStack stack = lane_stack(p.scene, s_stack + threadIdx.x, kBlockSize);
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntWalk + 32u * queue], p.capacity);
  if (count == 0u)
    return;
  const BvhNodes nodes = stage_nodes(scene, s_nodes, kWalkLdsNodes);  // every event descends from the root: its first levels come from LDS
  const uint32_t mode = bdpt_mode(it);
  const bool use_mis = opt_enable_mis(it);
  BdptState st = {};
  BdptWalk walk = {kInvalid, kInvalid, 0u};
  bool active = false, exhausted = false;
  uint32_t entry = kInvalid, budget = 0u;
  while (walk_refill(p, count, active, exhausted, entry)) {
    if (entry != kInvalid) {
      st = bdpt_load(p.walk[queue], entry);
      st.prev.tri = st.prev_slot;
      walk = bdpt_walk_info(p, queue, entry);
      budget = kWalkBudget;
      active = true;
    }
    BdptCameraStep r = {};
    float4 h = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInvalid));
    uint32_t next = 0u;
    if (active) {
      r = bdpt_camera_step<kStepWalkEvent, true>(p, scene, nodes, stack, it, mode, use_mis, st, walk, h);
      next = bdpt_camera_finish<kStepWalkEvent>(p, scene, st, walk, r, 0u);
    }
    const uint32_t exit_slot = wave_compact_slot(r.exit, p.counters + kCntWalkExit);
    if (r.exit && (exit_slot < p.capacity)) {
      bdpt_store(p.walk_exit, exit_slot, st, st.prev.tri);
      p.walk_exit_hits[exit_slot] = h;
    }
    active = next == 2u;
    budget -= active ? 1u : 0u;
    const bool later = active && (budget == 0u);  // out of budget: the walk continues in the next round
    const uint32_t later_slot = wave_compact_slot(later, p.counters + kCntWalk + 32u * (queue ^ 1u));
    if (later)
      bdpt_walk_push(p, queue ^ 1u, later_slot, st, st.prev.tri, walk);
    active = active && (later == false);
  }
}

template <bool kSimple>
__global__ __launch_bounds__(kBlockSize) void k_bdpt_walk_exit_camera(Pipeline p, VcmParams it, uint32_t out_set) {
  __shared__ BlockScratch s_scratch;
  const LaneStack no_stack = {};
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntWalkExit], p.capacity);
  const PathSet& out = p.paths[out_set];
  uint32_t* out_counter = p.counters + (out_set == 0 ? kCntActiveA : kCntActiveB);
  const uint32_t mode = bdpt_mode(it);
  const bool use_mis = opt_enable_mis(it);
  ETX_BLOCK_LOOP(count, i) {
    const bool valid = i < count;
    BdptState st = {};
    BdptWalk walk = {kInvalid, kInvalid, 0u};
    BdptCameraStep r = {};
    if (valid) {
      st = bdpt_load(p.walk_exit, i);
      st.prev.tri = st.prev_slot;
      float4 h = p.walk_exit_hits[i];
      r = bdpt_camera_step<kStepWalkExit, kSimple>(p, scene, global_nodes(scene), no_stack, it, mode, use_mis, st, walk, h);
    }
    const uint32_t vertex_slot = block_compact_slot(r.store_vertex, p.counters + kCntCameraVertices, s_scratch);
    const uint32_t next = valid ? bdpt_camera_finish<kStepWalkExit>(p, scene, st, walk, r, vertex_slot) : 0u;
    const uint32_t slot = block_compact_slot(next == 1u, out_counter, s_scratch);
    if (next == 1u)
      bdpt_store(out, slot, st, st.prev.tri);
  }
}

// connect_camera_to_light (:1342-1378) with mis_weight_camera_to_light (:1079-1133) for the camera vertices of this bounce
template <bool kSimple, uint32_t kPart>
__global__ __launch_bounds__(kBlockSize) void k_bdpt_connect_light(Pipeline p, VcmParams it) {
  static_assert((kPart == kPartAll) || (kSimple == (kPart == kPartSimple)), "part / instantiation");
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  const uint32_t count = (kPart == kPartGeneral) ? min(p.counters[kCntGroupSubsurface], p.capacity) : min(p.counters[kCntCameraVertices], p.cv_capacity);
  const uint32_t mode = bdpt_mode(it);
  ETX_BLOCK_LOOP(count, j) {
    ShadowRequest request;
    bool queue = false;
    bool valid = (j < count) && opt_connect_to_light(it);
    const uint32_t i = (kPart == kPartGeneral) ? (valid ? p.group_list[1][j] : 0u) : j;
    if (kPart == kPartSimple)  // (the class bit of a camera vertex record: bdpt_store_camera_vertex)
      valid = bdpt_bin_general(p, 1u, i, valid, valid && ((__float_as_uint(p.cv.thr_depth[i].w) & kCvGeneralBsdfBit) != 0u), s_scratch);
    if (valid) {
      BdptCameraVertex z = bdpt_load_camera_vertex(p, scene, i);
      const uint32_t connection_len = z.path_size;  // camera_path_length() + 1
      if ((connection_len <= scene.max_path_length) && (connection_len >= scene.min_path_length)) {
        Sampler smp;
        smp.seed = z.seed, smp.fixed_u = smp.fixed_v = smp.fixed_w = 0.0f;
        const uint32_t emitter_index = sample_emitter_index(scene, z.rnd_fixed.z);
        const EmitterSample es = sample_emitter(scene, emitter_index, f2{z.rnd_fixed.x, z.rnd_fixed.y}, z.full.isect.pos, z.wavelength);
        const f3 dp = es.origin - z.full.isect.pos;
        if ((is_zero(es.value) == false) && (es.pdf_dir != 0.0f) && (dot(dp, dp) > kEpsilon)) {
          const BdptBsdf bsdf = bdpt_bsdf<kSimple>(scene, z.full, kPathCamera, es.direction, z.wavelength, smp);
          if (is_zero(bsdf.bsdf) == false) {
            const float sampling_pdf = es.pdf_dir * es.pdf_sample;
            float weight = 1.0f;
            if (opt_enable_mis(it)) {
              if (mode == kBdptPathTracing) {
                weight = power_heuristic(sampling_pdf, es.is_delta ? 0.0f : bsdf.pdf);
              } else {
                const etx_abi_emitter& em = scene.emitters[es.emitter_index];
                const BVtx sampled = {es.origin, es.normal, 0.0f, 0.0f, kBvEmitter | ((es.triangle_index != kInvalid) ? kBvSurface : 0u), es.triangle_index};
                const BVtx z_curr = z.full.summary(z.from_prev);
                const float p_sample = bdpt_emitter_sample_pdf(scene, em, es.direction);
                const float from_emitter = bdpt_pdf_from_emitter(scene, es.emitter_index, es.origin, es.normal, z_curr);
                const float z_prev_backward = bdpt_pdf_area<kSimple>(scene, kPathLight, es.origin, z.full, z.prev, z.wavelength, smp);
                const float p_bsdf_sample = bdpt_pdf_area<kSimple>(scene, kPathCamera, z.prev.pos, z.full, sampled, z.wavelength, smp);
                if (mode == kBdptFast) {  // :1114-1129
                  const float p_fwd = z.prev.from_prev * z.from_prev;
                  const float p_connection = p_fwd * p_sample;
                  const float p_direct = es.is_delta ? 0.0f : p_fwd * p_bsdf_sample;
                  const float p_bck = z.prev.history * ((z.path_size > 2u) ? z_prev_backward : 1.0f);
                  weight = balance_heuristic(p_connection, p_direct, p_sample * from_emitter * p_bck);
                } else {
                  const float w_camera = bdpt_mis_camera(z.path_size, from_emitter, z.from_prev, z_prev_backward, z.prev);
                  const float w_light = es.is_delta ? 0.0f : safe_div(p_bsdf_sample, p_sample);
                  weight = 1.0f / (w_camera + 1.0f + w_light);
                }
              }
            }
            const f3 value = z.throughput * bsdf.bsdf * (es.value / sampling_pdf) * weight * spectral_film_weight(scene, z.wavelength);
            request = {bdpt_segment_origin(scene, z.full, es.origin), es.origin, value, z.medium, film_index(it, z.pixel), z.wavelength};
            queue = true;
          }
        }
      }
    }
    const uint32_t slot = block_compact_slot(queue, p.counters + kCntShadow, s_scratch);
    if (queue)
      write_shadow(p, slot, request);
  }
}

// connect_camera_to_light_path (:438-497), one (camera vertex, light vertex) pair per lane, with
// mis_weight_camera_to_light_path (:1184-1209)
// Whether a listed pair takes part at all (k_bdpt_expand_pairs lists every vertex of the light path but the emitter's; the class of the pair is the list it is on)
ETX_DEV bool bdpt_pair_connects(const Pipeline& p, const DScene& scene, const uint2 pair) {
  if ((pair.x >= p.cv_capacity) || (pair.y >= p.lv.capacity))
    return false;  // a slot of an overflowed (discarded) iteration's lists that nobody wrote: whatever the freshly grown buffer held
  const uint32_t y_flags = __float_as_uint(p.lv.thr_dvm(pair.y).w);
  const uint32_t light_s = __float_as_uint(p.lv.bc_len_med(pair.y).z) >> 16u;
  const uint32_t z_word = __float_as_uint(p.cv.thr_depth[pair.x].w);
  const uint32_t camera_path_size = z_word & ~(kCvExitMaterialBit | kCvGeneralBsdfBit);
  const uint32_t target_path_length = (camera_path_size - 1u) + light_s + 1u;
  return (light_s >= 1u) && ((y_flags & kBvConnectible) != 0u) && (target_path_length >= scene.min_path_length) && (target_path_length <= scene.max_path_length);
}

// connect_camera_to_light_path for one (camera vertex, light vertex) pair: the visibility request, or false
template <bool kSimple>
ETX_DEV bool bdpt_connect_pair(const Pipeline& p, const DScene& scene, const VcmParams& it, const uint2 pair, ShadowRequest& request) {
  const BdptCameraVertex z = bdpt_load_camera_vertex(p, scene, pair.x);
  const BdptLightVertex y = bdpt_load_light_vertex(p, scene, pair.y);
  f3 dw = z.full.isect.pos - y.self.pos;
  const float dwl = dot(dw, dw);
  if ((dwl > kInvMaxHalf) == false)
    return false;
  dw = dw * (1.0f / sqrtf(dwl));
  Sampler smp;
  // one stream per (camera vertex, light vertex), keyed by the two vertices' own seeds - not by the light vertex's pool slot, which differs from run to run
  smp.seed = Sampler::random_seed(z.seed, y.seed ^ y.index_in_path), smp.fixed_u = smp.fixed_v = smp.fixed_w = 0.0f;
  const f3 bsdf_y = bdpt_bsdf<kSimple>(scene, y.full, kPathLight, dw, z.wavelength, smp).bsdf;
  const f3 bsdf_z = bdpt_bsdf<kSimple>(scene, z.full, kPathCamera, -dw, z.wavelength, smp).bsdf;
  const f3 connect = y.throughput * bsdf_y * bsdf_z;
  if (is_zero(connect))
    return false;
  float weight = 1.0f;
  if (opt_enable_mis(it)) {
    const BVtx y_prev = bdpt_load_light_summary(p, y.prev);
    const BVtx z_curr = z.full.summary(z.from_prev);
    const float z_curr_pdf = bdpt_pdf_area<kSimple>(scene, kPathLight, y_prev.pos, y.full, z_curr, z.wavelength, smp);
    const float z_prev_pdf = bdpt_pdf_area<kSimple>(scene, kPathCamera, y.self.pos, z.full, z.prev, z.wavelength, smp);
    const float y_curr_pdf = bdpt_pdf_area<kSimple>(scene, kPathCamera, z.prev.pos, z.full, y.self, z.wavelength, smp);
    const float y_prev_pdf = bdpt_pdf_area<kSimple>(scene, kPathLight, z.full.isect.pos, y.full, y_prev, z.wavelength, smp);
    const float w_camera = bdpt_mis_camera(z.path_size, z_curr_pdf, z.from_prev, z_prev_pdf, z.prev);
    const float w_light = bdpt_mis_light(y_curr_pdf, y.self.from_prev, y_prev_pdf, y_prev);
    weight = 1.0f / (1.0f + w_camera + w_light);
  }
  const f3 value = connect * z.throughput * (weight / dwl) * spectral_film_weight(scene, z.wavelength);
  request = {bdpt_segment_origin(scene, y.full, z.full.isect.pos), z.full.isect.pos, value, y.medium, film_index(it, z.pixel), z.wavelength};
  return true;
}

// (camera vertex, light path of its pixel) -> (camera vertex, light vertex) pairs: connect_camera_to_light_path (:438-497) flattened, as k_expand_pairs does for VCM
// (kernels_connect.hip), with two differences. TWO dense lists in the one pair buffer: pairs of two vertices of simple BSDF classes from the front (kCntPairs), pairs
// with a vertex of a general class from the BACK (kCntPairsGeneral; entry j at pair_capacity - 1 - j) - until round 6 both pair kernels streamed over one list and read
// three words of the two records of EVERY pair to find their own (the general kernel: 9 ms per step of configs[3] for the few pairs it evaluates). The class of a light
// vertex is a bit of its list entry, a path's count of them is in its row header, so the lists' sizes are known before anything is read. And the path's vertices come
// from its row and its CHUNKS (pipeline.h kBdptRowHeader) with independent 16-byte loads - no walk along the vertices' links (9 ms per step: one lane chasing hundreds
// of pointers while its workgroup waits). The emitter's vertex (entry 0) connects to nothing (light_s >= 1, :447) and is not listed.
__global__ __launch_bounds__(kBlockSize) void k_bdpt_expand_pairs(Pipeline p, VcmParams it) {
  __shared__ uint32_t s_wave_total[2][kBlockSize / 64u];
  __shared__ uint32_t s_base[2];
  const uint32_t count = min(p.counters[kCntCameraVertices], p.cv_capacity);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6u;
  const uint32_t row_entries = p.path_table_entries - kBdptRowHeader;
  ETX_BLOCK_LOOP(count, i) {
    uint32_t length = 0u, general = 0u, path = 0u;
    bool camera_general = false;
    uint4 header = make_uint4(kInvalid, 0u, 0u, 0u);  // newest chunk, length, vertices of general classes, entry 0: one 16-byte load of the path's row
    if (i < count) {
      path = __float_as_uint(p.cv.mis_pixel[i].w);
      camera_general = (__float_as_uint(p.cv.thr_depth[i].w) & kCvGeneralBsdfBit) != 0u;
      header = p.light_path_table[size_t(path) * (p.path_table_entries >> 2u)];
      length = header.y;
    }
    const uint32_t k = (length > 1u) ? (length - 1u) : 0u;  // without the emitter's vertex
    const uint32_t kg = camera_general ? k : min(header.z, k);
    const uint32_t ks = k - kg;
    // workgroup exclusive prefix sums of the two counts: wave scans, then one reservation per list for all four waves
    uint32_t incl_s = ks, incl_g = kg;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t ts = __shfl_up(incl_s, d), tg = __shfl_up(incl_g, d);
      if (lane >= d)
        incl_s += ts, incl_g += tg;
    }
    if (lane == 63u)
      s_wave_total[0][wave] = incl_s, s_wave_total[1][wave] = incl_g;
    __syncthreads();
    if (threadIdx.x == 0u) {
      uint32_t total_s = 0u, total_g = 0u;
#pragma unroll
      for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
        total_s += s_wave_total[0][w], total_g += s_wave_total[1][w];
      s_base[0] = total_s ? atomicAdd(p.counters + kCntPairs, total_s) : 0u;
      s_base[1] = total_g ? atomicAdd(p.counters + kCntPairsGeneral, total_g) : 0u;
      if (total_s + total_g)
        atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatPairs), (unsigned long long)(total_s + total_g));
    }
    __syncthreads();
    uint32_t base_s = s_base[0] + incl_s - ks, base_g = s_base[1] + incl_g - kg;
#pragma unroll
    for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
      base_s += (w < wave) ? s_wave_total[0][w] : 0u, base_g += (w < wave) ? s_wave_total[1][w] : 0u;
    __syncthreads();
    if (k == 0u)
      continue;
    if ((base_s + ks > p.pair_capacity) || (base_g + kg > p.pair_capacity)) {  // the buffer's bounds; whether the two lists MET is the pair kernel's check (both counts are final there)
      atomicOr(p.counters + kCntOverflow, kOverflowPairs);
      continue;
    }
    uint32_t at_s = base_s, at_g = p.pair_capacity - 1u - base_g;
    auto put = [&](uint32_t entry) {
      const uint2 pair = make_uint2(i, entry & ~kPathEntryGeneralBit);
      if (camera_general || ((entry & kPathEntryGeneralBit) != 0u)) {
        if (at_g + kg > p.pair_capacity - 1u - base_g)  // (a row whose class count and entries disagree - an iteration that is being discarded - must not leave its range)
          p.pairs[at_g--] = pair;
      } else if (at_s < base_s + ks) {
        p.pairs[at_s++] = pair;
      }
    };
    // entries 1 .. of the row: word kBdptRowHeader + j of the row = entry j
    const uint4* row = p.light_path_table + size_t(path) * (p.path_table_entries >> 2u);
    const uint32_t from_row = min(length, row_entries);
    for (uint32_t q = 1; (q << 2u) < from_row + kBdptRowHeader; ++q) {
      const uint4 t = row[q];
      const uint32_t j = (q << 2u) - kBdptRowHeader;  // 1, 5, 9, ...
      put(t.x);
      if (j + 1u < from_row)
        put(t.y);
      if (j + 2u < from_row)
        put(t.z);
      if (j + 3u < from_row)
        put(t.w);
    }
    // ... and of the chunks, newest first: [0] the previous chunk, then kPathChunkEntries entries (all but the newest are full)
    if (length > row_entries) {
      uint32_t remaining = length - row_entries;
      uint32_t in_chunk = ((remaining - 1u) % kPathChunkEntries) + 1u;
      uint32_t chunk = header.x;
      while ((remaining != 0u) && (chunk < p.path_chunk_capacity)) {
        const uint4* c = reinterpret_cast<const uint4*>(p.path_chunks + size_t(chunk) * kPathChunkWords);
        const uint4 t0 = c[0];
        put(t0.y);
        if (in_chunk > 1u)
          put(t0.z);
        if (in_chunk > 2u)
          put(t0.w);
        for (uint32_t q = 1; (q << 2u) - 1u < in_chunk; ++q) {
          const uint4 t = c[q];
          const uint32_t position = (q << 2u) - 1u;  // 3, 7, 11, ...
          put(t.x);
          if (position + 1u < in_chunk)
            put(t.y);
          if (position + 2u < in_chunk)
            put(t.z);
          if (position + 3u < in_chunk)
            put(t.w);
        }
        remaining -= in_chunk;
        in_chunk = kPathChunkEntries;
        chunk = t0.x;
      }
    }
  }
}

// One pair per lane over one of the two lists: kGeneral = false the pairs of two vertices of simple classes (inline Lambert / phase-function code; every pair of a
// scene of simple materials), true the pairs with a vertex of a general class (out-of-line BSDF library) - dense, so every lane of the register-heavy code has a pair.
template <bool kGeneral>
__global__ __launch_bounds__(kBlockSize) void k_bdpt_connect_pairs(Pipeline p, VcmParams it) {
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  const uint32_t front = p.counters[kCntPairs], back = p.counters[kCntPairsGeneral];
  if ((kGeneral == false) && (blockIdx.x == 0u) && (threadIdx.x == 0u) && ((front > p.pair_capacity) || (back > p.pair_capacity) || (front + back > p.pair_capacity)))
    atomicOr(p.counters + kCntOverflow, kOverflowPairs);  // the two lists met: the iteration is discarded, the pool grows, it is rendered again
  const uint32_t count = min(kGeneral ? back : front, p.pair_capacity);
  ETX_BLOCK_LOOP(count, i) {
    ShadowRequest request;
    bool queue = false;
    if (i < count) {
      const uint2 pair = p.pairs[kGeneral ? (p.pair_capacity - 1u - i) : i];
      if (bdpt_pair_connects(p, scene, pair))
        queue = bdpt_connect_pair<kGeneral == false>(p, scene, it, pair, request);
    }
    const uint32_t slot = block_compact_slot(queue, p.counters + kCntShadow, s_scratch);
    if (queue)
      write_shadow(p, slot, request);
  }
}

// ---------------------------------------------------------------------------------------------------------------
void launch_bdpt_light_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it) {
  hipLaunchKernelGGL(k_bdpt_light_generate, dim3(grid_for(p.capacity)), dim3(kBlockSize), 0, stream, p, it);
}
// `variant` (kernels.h kBdptKernels*): Simple = every material in use is of a class the inline Lambert / delta BSDFs answer for (DeviceScene::simple_materials);
// General = every class through the out-of-line dispatch (dev_bsdf_ool.h) for every item; Binned = a mixed scene whose subsurface scatter material is of a simple
// class: the inline instantiation over every item, the out-of-line one over the items it handed over (kPartSimple / kPartGeneral above)
#define ETX_BDPT_LAUNCH(KERNEL, GRID, ...)                                                                                    \
  do {                                                                                                                        \
    if (variant == kBdptKernelsSimple) {                                                                                      \
      hipLaunchKernelGGL((KERNEL<true, kPartAll>), dim3(GRID), dim3(kBlockSize), 0, stream, __VA_ARGS__);                     \
    } else if (variant == kBdptKernelsBinned) {                                                                               \
      hipLaunchKernelGGL((KERNEL<true, kPartSimple>), dim3(GRID), dim3(kBlockSize), 0, stream, __VA_ARGS__);                  \
      hipLaunchKernelGGL((KERNEL<false, kPartGeneral>), dim3(GRID), dim3(kBlockSize), 0, stream, __VA_ARGS__);                \
    } else {                                                                                                                  \
      hipLaunchKernelGGL((KERNEL<false, kPartAll>), dim3(GRID), dim3(kBlockSize), 0, stream, __VA_ARGS__);                    \
    }                                                                                                                         \
  } while (0)
// the exit vertex of a walk carries the scatter material (build_path :858-861): of a simple class in a binned scene
#define ETX_BDPT_LAUNCH_EXIT(KERNEL, GRID, ...)                                                             \
  do {                                                                                                      \
    if (variant != kBdptKernelsGeneral)                                                                     \
      hipLaunchKernelGGL(KERNEL<true>, dim3(GRID), dim3(kBlockSize), 0, stream, __VA_ARGS__);               \
    else                                                                                                    \
      hipLaunchKernelGGL(KERNEL<false>, dim3(GRID), dim3(kBlockSize), 0, stream, __VA_ARGS__);              \
  } while (0)

void launch_bdpt_light_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, uint32_t variant) {
  ETX_BDPT_LAUNCH(k_bdpt_light_shade, max(1u, grid_for(min(p.capacity, max_items))), p, it, in_set);
}
// the walks of the paths the shade kernel of this round put on the walk queue: the scattering events in persistent wavefronts (at most
// kWalkBlocks workgroups: 32 KB of traversal stack each), then the exit vertices as a dense kernel
void launch_bdpt_walk(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool camera, uint32_t in_set, uint32_t max_items, uint32_t variant) {
  const uint32_t items = min(p.capacity, max_items);
  const uint32_t blocks = max(1u, min(kWalkBlocks, (items + kBlockSize - 1u) / kBlockSize));
  if (camera) {
    hipLaunchKernelGGL(k_bdpt_walk_camera, dim3(blocks), dim3(kBlockSize), 0, stream, p, it, in_set);
    ETX_BDPT_LAUNCH_EXIT(k_bdpt_walk_exit_camera, max(1u, grid_for(items)), p, it, in_set ^ 1u);
  } else {
    hipLaunchKernelGGL(k_bdpt_walk_light, dim3(blocks), dim3(kBlockSize), 0, stream, p, it, in_set);
    ETX_BDPT_LAUNCH_EXIT(k_bdpt_walk_exit_light, max(1u, grid_for(items)), p, it, in_set ^ 1u);
  }
}
void launch_bdpt_connect_camera(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items, uint32_t variant) {
  ETX_BDPT_LAUNCH(k_bdpt_connect_camera, max(1u, grid_for(uint32_t(min(uint64_t(max_items) * 2ull, uint64_t(p.lv.capacity))))), p, it);
}
void launch_bdpt_camera_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it) {
  hipLaunchKernelGGL(k_bdpt_camera_generate, dim3(grid_for(p.capacity)), dim3(kBlockSize), 0, stream, p, it);
}
void launch_bdpt_camera_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, uint32_t variant) {
  ETX_BDPT_LAUNCH(k_bdpt_camera_shade, max(1u, grid_for(min(p.capacity, max_items))), p, it, in_set);
}
void launch_bdpt_connect_light(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items, uint32_t variant) {
  ETX_BDPT_LAUNCH(k_bdpt_connect_light, max(1u, grid_for(min(p.capacity, max_items))), p, it);
}
// Vertex connections: the expansion into the two pair lists, then one kernel per list (the general one only where a general class exists)
void launch_bdpt_expand_pairs(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items) {
  hipLaunchKernelGGL(k_bdpt_expand_pairs, dim3(max(1u, grid_for(min(max_items, p.capacity)))), dim3(kBlockSize), 0, stream, p, it);
}
void launch_bdpt_connect_pairs(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items, uint32_t variant) {
  const uint32_t pair_blocks = max(1u, grid_for(uint32_t(min(uint64_t(max_items) * 8ull, uint64_t(p.pair_capacity)))));
  hipLaunchKernelGGL(k_bdpt_connect_pairs<false>, dim3(pair_blocks), dim3(kBlockSize), 0, stream, p, it);
  if (variant != kBdptKernelsSimple)
    hipLaunchKernelGGL(k_bdpt_connect_pairs<true>, dim3(pair_blocks), dim3(kBlockSize), 0, stream, p, it);
}

}  // namespace etxd
