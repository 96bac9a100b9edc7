// kernels_vcm.hip - VCM wavefront kernels (light pass, camera pass, connections, photon merge) for gfx950.
//
// Kernel decomposition (one iteration, see host_api.cpp for the launch sequence):
//   k_iteration_reset                       counters, light path heads
//   k_light_generate                        vcm_generate_emitter_state          (vcm_shared.hxx:310-349)
//   loop: k_trace_closest ; k_light_shade   vcm_light_step                      (vcm_shared.hxx:1090-1260)
//   k_grid_* (kernels_grid.hip)             VCMSpatialGrid::construct           (vcm_shared.cxx:49-152)
//   k_camera_generate                       vcm_generate_camera_state           (vcm_shared.hxx:351-377)
//   loop: k_trace_closest ; k_camera_shade ; k_connect ; k_merge
//                                           vcm_camera_step                     (vcm_shared.hxx:927-1079)
// All kernels: 256-thread blocks, persistent grid, wave-uniform grid-stride loops, survivors compacted with a wave
// ballot + one atomic per wavefront. Shading is fp32 VALU / divergence bound (no MFMA: there is no contraction).
#include "kernels_shade.inl"  // k_light_shade / k_camera_shade templates; the <false> instantiations are separate translation units

#include <mutex>
#include <unordered_map>
#include "tuning_knobs.h"

namespace etxd {

uint32_t resident_blocks(const void* kernel) {
  static std::mutex guard;
  static std::unordered_map<const void*, uint32_t> cache;
  std::lock_guard<std::mutex> lock(guard);
  auto found = cache.find(kernel);
  if (found != cache.end())
    return found->second;
  int per_cu = 0, device = 0, cus = 0;
  uint32_t value = kPersistentBlocks;
  if ((hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, int(kBlockSize), 0) == hipSuccess) && (hipGetDevice(&device) == hipSuccess) &&
      (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) && (per_cu > 0) && (cus > 0))
    value = uint32_t(per_cu) * uint32_t(cus);
  cache.emplace(kernel, value);
  return value;
}

uint32_t persistent_grid(uint32_t blocks_needed, const void* kernel, uint32_t percent) {
  uint32_t limit = kPersistentBlocks;
  if (percent != 0u)
    limit = uint32_t(min(uint64_t(kPersistentBlocks), max(uint64_t(64), uint64_t(resident_blocks(kernel)) * percent / 100u)));
  return max(1u, min(blocks_needed, limit));
}


#define ETX_WAVE_LOOP(COUNT)                                                          \
  const uint32_t lane_ = threadIdx.x & 63u;                                            \
  const uint32_t stride_ = gridDim.x * blockDim.x;                                     \
  for (uint32_t base_ = blockIdx.x * blockDim.x + threadIdx.x - lane_; base_ < (COUNT); base_ += stride_)

static uint32_t grid_for(uint32_t capacity) {
  return min(kPersistentBlocks, (capacity + kBlockSize - 1) / kBlockSize);
}

ETX_DEV uint32_t float_to_ordered(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockSize) void k_iteration_reset(Pipeline p) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t stride = gridDim.x * blockDim.x;
  if (tid < kCounterCount) {
    uint32_t v = 0u;
    if ((tid >= kCntBboxMin) && (tid < kCntBboxMin + 3))
      v = 0xffffffffu;
    p.counters[tid] = v;
  }
  if (tid < kBlockStatRows * kBlockStatCount)
    p.block_stats[tid] = 0ull;
  for (uint32_t i = tid; i < p.capacity; i += stride)  // head (bidirectional: newest chunk) = none, length 0, (bidirectional: vertices of general classes) 0: the header of every path's table row
    p.light_path_table[size_t(i) * (p.path_table_entries >> 2u)] = make_uint4(kInvalid, 0u, 0u, 0u);
}

void launch_iteration_reset(hipStream_t stream, const Pipeline& p) {
  hipLaunchKernelGGL(k_iteration_reset, dim3(grid_for(p.capacity)), dim3(kBlockSize), 0, stream, p);
}

// ---------------------------------------------------------------------------------------------------------------
// vcm_generate_emitter_state, vcm_shared.hxx:310-349 (one light path per pixel index, vcm_cpu.cxx:139-140)
__global__ __launch_bounds__(kBlockSize) void k_light_generate(Pipeline p, VcmParams it) {
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  ETX_BLOCK_LOOP(it.path_count, i) {
    bool valid = false;
    PathState st;
    if (i < it.path_count) {
      st.sampler.init(i, it.iteration);
      st.id = i;
      // vcm_shared.hxx:313: spectral scenes draw the wavelength first
      st.wavelength = scene.spectral ? spectral_sample_wavelength(st.sampler.next()) : 0.0f;
      p.path_wavelength[i] = st.wavelength;
      EmitterSample es = sample_emission(scene, st.sampler, st.wavelength);
      if (es.pdf_dir > 0.0f) {
        float cos_t = dot(es.direction, es.normal);
        st.throughput = es.value * (cos_t / (es.pdf_dir * es.pdf_area * es.pdf_sample));
        st.ray_o = es.origin;
        st.ray_d = es.direction;
        st.ray_tmin = kRayEpsilon;
        st.ray_tmax = kMaxFloat;
        if (es.triangle_index != kInvalid)
          st.ray_o = shading_pos(scene, scene.triangles[es.triangle_index], es.barycentric, st.ray_d);
        st.d_vcm = es.is_distant ? 1.0f / es.pdf_area : 1.0f / es.pdf_dir;
        st.d_vc = es.is_delta ? 0.0f : (es.is_distant ? 1.0f : cos_t) / (es.pdf_dir * es.pdf_area * es.pdf_sample);
        st.d_vm = st.d_vc * it.vc_weight;
        st.path_distance = 0.0f;
        st.eta = 1.0f;
        st.medium = es.medium_index;
        st.depth = 0u;
        st.flags = (es.is_delta ? kPathDeltaEmitter : 0u) | (es.is_distant ? 0u : kPathLocalEmitter);
        valid = true;
      }
    }
    uint32_t slot = block_compact_slot(valid, p.counters + kCntActiveA, s_scratch);
    if (valid)
      store_path(p.paths[0], slot, st);
  }
}

void launch_light_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it) {
  hipLaunchKernelGGL(k_light_generate, dim3(grid_for(p.capacity)), dim3(kBlockSize), 0, stream, p, it);
}

void launch_light_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups) {
  const dim3 grid(max(1u, grid_for(min(p.capacity, max_items))));
  // the simple group's kernel stages the small tables in LDS once per workgroup (dev_stage.h) and runs two workgroups per CU: no more workgroups
  // than are resident at once, each loops longer
  static const uint32_t percent = etxh::tuning_knob("ETX_HIP_GRID_SHADE", 100u);
  const void* simple_kernel = groups.binned() ? reinterpret_cast<const void*>(&k_light_shade<kShadeGroupSimple, true>) : reinterpret_cast<const void*>(&k_light_shade<kShadeGroupSimple, false>);
  const dim3 simple_grid(persistent_grid(grid.x, simple_kernel, percent));
  if (groups.binned() == false) {
    hipLaunchKernelGGL((k_light_shade<kShadeGroupSimple, false>), simple_grid, dim3(kBlockSize), 0, stream, p, it, in_set);
    return;
  }
  hipLaunchKernelGGL((k_light_shade<kShadeGroupSimple, true>), simple_grid, dim3(kBlockSize), 0, stream, p, it, in_set);
  if (groups.general)
    launch_light_shade_group(stream, p, it, in_set, grid, kShadeGroupGeneral);
  if (groups.subsurface)
    launch_light_shade_group(stream, p, it, in_set, grid, kShadeGroupSubsurface);
}

// ---------------------------------------------------------------------------------------------------------------
// vcm_generate_camera_state, vcm_shared.hxx:351-377 (all pixels active: Film::active_pixel with pixel_size 1)
__global__ __launch_bounds__(kBlockSize) void k_camera_generate(Pipeline p, VcmParams it) {
  const DScene& scene = p.scene;
  ETX_WAVE_LOOP(it.path_count) {
    const uint32_t i = base_ + lane_;
    if (i >= it.path_count)
      continue;
    PathState st;
    st.id = i;
    // The reference seeds camera path i exactly like light path i (vcm_shared.hxx:312,357). Its two streams drift
    // apart only because alpha_test_pass draws one number per BVH candidate inside rt.trace (scene_bsdf.hxx:143);
    // the device traversal draws none for opaque triangles, so with the same seed camera draw k+1 would equal
    // light draw k forever and the (pixel i, light path i) vertex connections become correlated -> biased
    // (measured: -7% in the connection-only image). The camera stream is therefore re-keyed.
    // etx_abi_vcm_options::reference_seeding ("hip-reference_seeding") keeps the shared seed: the state the reference is in when its
    // candidate draws are taken off the path's stream (oracle shim, ETX_ORACLE_BVH_DRAWS=opaque_none) - tests/test_gpu_parity_hi.py and
    // tests/test_gpu_options.py compare the two.
    st.sampler.init(i, it.iteration);
    if ((it.options & kOptionReferenceSeeding) == 0u)
      st.sampler.seed = Sampler::random_seed(st.sampler.seed, 0x43414d45u);
    // vcm_shared.hxx:358-359: one draw is consumed, the wavelength is the one of light path i (vcm_cpu.cxx:186)
    st.wavelength = 0.0f;
    if (scene.spectral) {
      (void)st.sampler.next();
      st.wavelength = p.path_wavelength[i];
    }
    uint32_t px = i % it.film_w, py = i / it.film_w;
    f2 uv = get_jittered_uv(st.sampler, px, py, it.film_w, it.film_h);
    RayGen r = generate_ray(scene, uv, st.sampler.next_2d());
    st.ray_o = r.o, st.ray_d = r.d, st.ray_tmin = r.tmin, st.ray_tmax = r.tmax;
    st.throughput = mk3(1.0f);
    st.d_vcm = 1.0f / film_evaluate_out_pdf_dir(scene, st.ray_d);
    st.d_vc = 0.0f, st.d_vm = 0.0f;
    st.medium = scene.camera.medium_index;
    st.eta = 1.0f;
    st.path_distance = 0.0f;
    st.depth = 1u;
    st.flags = 0u;
    store_path(p.paths[0], i, st);
  }
  if ((blockIdx.x == 0) && (threadIdx.x == 0))
    p.counters[kCntActiveA] = it.path_count;
}

void launch_camera_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it) {
  hipLaunchKernelGGL(k_camera_generate, dim3(grid_for(p.capacity)), dim3(kBlockSize), 0, stream, p, it);
}


void launch_camera_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups) {
  const dim3 grid(max(1u, grid_for(min(p.capacity, max_items))));
  // the simple group's kernel stages the small tables in LDS once per workgroup (dev_stage.h) and runs two workgroups per CU: no more workgroups
  // than are resident at once, each loops longer
  static const uint32_t percent = etxh::tuning_knob("ETX_HIP_GRID_SHADE", 100u);
  const void* simple_kernel = groups.binned() ? reinterpret_cast<const void*>(&k_camera_shade<kShadeGroupSimple, true>) : reinterpret_cast<const void*>(&k_camera_shade<kShadeGroupSimple, false>);
  const dim3 simple_grid(persistent_grid(grid.x, simple_kernel, percent));
  if (groups.binned() == false) {
    hipLaunchKernelGGL((k_camera_shade<kShadeGroupSimple, false>), simple_grid, dim3(kBlockSize), 0, stream, p, it, in_set);
    return;
  }
  hipLaunchKernelGGL((k_camera_shade<kShadeGroupSimple, true>), simple_grid, dim3(kBlockSize), 0, stream, p, it, in_set);
  if (groups.general)
    launch_camera_shade_group(stream, p, it, in_set, grid, kShadeGroupGeneral);
  if (groups.subsurface)
    launch_camera_shade_group(stream, p, it, in_set, grid, kShadeGroupSubsurface);
}

#if defined(ETX_HIP_DEBUG_COUNTERS)
__global__ void k_debug_lists(Pipeline p) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.capacity)
    return;
  uint32_t len = 0;
  for (uint32_t vi = reinterpret_cast<const uint32_t*>(p.light_path_table)[size_t(i) * p.path_table_entries]; (vi != kInvalid) && (len < 100000u); vi = p.lv.next(vi))
    len++;
  atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kDbgBase + 2), (unsigned long long)len);
  atomicMax(p.counters + kDbgBase + 4, len);
}
void launch_debug_lists(hipStream_t stream, const Pipeline& p) {
  hipLaunchKernelGGL(k_debug_lists, dim3((p.capacity + 255u) / 256u), dim3(256), 0, stream, p);
}
#endif

// Folds the per-workgroup statistics rows into the u64 counters the host reads (once per iteration).
__global__ __launch_bounds__(kBlockSize) void k_stats_finalize(Pipeline p) {
  __shared__ unsigned long long s_sum[kBlockStatCount];
  if (threadIdx.x < kBlockStatCount)
    s_sum[threadIdx.x] = 0ull;
  __syncthreads();
  for (uint32_t k = 0; k < kBlockStatCount; ++k) {
    unsigned long long v = 0ull;
    for (uint32_t row = threadIdx.x; row < kBlockStatRows; row += blockDim.x)
      v += p.block_stats[row * kBlockStatCount + k];
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1)
      v += __shfl_xor(v, d);
    if (((threadIdx.x & 63u) == 0u) && (v != 0ull))
      atomicAdd(&s_sum[k], v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *reinterpret_cast<unsigned long long*>(p.counters + kStatPhotonsExamined) = s_sum[kBlockStatExamined];
    *reinterpret_cast<unsigned long long*>(p.counters + kStatPhotonsMerged) = s_sum[kBlockStatMerged];
    *reinterpret_cast<unsigned long long*>(p.counters + kStatSplats) = s_sum[kBlockStatSplats];
    *reinterpret_cast<unsigned long long*>(p.counters + kStatCrossings) = s_sum[kBlockStatCrossings];
    *reinterpret_cast<unsigned long long*>(p.counters + kStatRaysExtension) += s_sum[kBlockStatCrossings];
    // the camera vertices the tail kernel (or the last bounce) left behind
    *reinterpret_cast<unsigned long long*>(p.counters + kStatCameraVertices) += (unsigned long long)p.counters[kCntCameraVertices];
  }
}

void launch_stats_finalize(hipStream_t stream, const Pipeline& p) {
  hipLaunchKernelGGL(k_stats_finalize, dim3(1), dim3(kBlockSize), 0, stream, p);
}

// ---------------------------------------------------------------------------------------------------------------
// End of a VCM iteration: the lane's iteration images go into the film sums (shared by the lanes: atomics) and are
// cleared for the next iteration. One add per pixel and iteration keeps the fp32 sums unbiased (host_api.cpp).
// An iteration whose pools overflowed is NOT committed: its images are cleared, the film stays as it was, and the host renders the same
// iteration again with larger pools (host_api.cpp execute_iteration) - the decision is taken here, on the device, from the overflow word.
__global__ __launch_bounds__(kBlockSize) void k_vcm_commit(float4* __restrict__ iteration_camera, float4* __restrict__ iteration_light, float4* __restrict__ camera_sum,
  float4* __restrict__ light_sum, uint32_t pixels, const uint32_t* __restrict__ counters) {
  const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const bool discard = counters[kCntOverflow] != 0u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += gridDim.x * blockDim.x) {
    if (discard) {
      iteration_camera[i] = zero, iteration_light[i] = zero;
      continue;
    }
    const float4 c = iteration_camera[i], l = iteration_light[i];
    atomicAdd(&camera_sum[i].w, 1.0f);  // iterations committed to this pixel (k_film_resolve, progressive read-back)
    if ((c.x != 0.0f) || (c.y != 0.0f) || (c.z != 0.0f)) {
      atomic_add_f3(camera_sum + i, f3{c.x, c.y, c.z});
      iteration_camera[i] = zero;
    }
    if ((l.x != 0.0f) || (l.y != 0.0f) || (l.z != 0.0f)) {
      atomic_add_f3(light_sum + i, f3{l.x, l.y, l.z});
      iteration_light[i] = zero;
    }
  }
}

void launch_vcm_commit(hipStream_t stream, float4* iteration_camera, float4* iteration_light, float4* camera_sum, float4* light_sum, uint32_t pixels, const uint32_t* counters) {
  hipLaunchKernelGGL(k_vcm_commit, dim3(grid_for(pixels)), dim3(kBlockSize), 0, stream, iteration_camera, iteration_light, camera_sum, light_sum, pixels, counters);
}

// ---------------------------------------------------------------------------------------------------------------
// Snapshot of the film sums for the multi-GPU reduce (host_reduce.h): the layers of `layer_mask` (bit = layer index in the film allocation) are
// copied, the others are left as they are in `snapshot` (zero since allocation). HBM-bound: 32 B per pixel and layer, 16-byte lanes.
// own_first / own_stride / film_w / film_h (own_stride > 1: a pixel-sharded bidirectional run): the commit of that integrator counts every pixel of the
// frame on every shard, so each shard contributes the count of the pixels it OWNS (pixel id = first + k * stride, pipeline.h path_pixel; the film is
// stored y-flipped) and zero for the others: the reduced count of a pixel is its owner's, whose camera values it normalises exactly whatever the
// other shards had committed when their snapshots were taken (until then shard 0's count stood for everyone's).
__global__ __launch_bounds__(kBlockSize) void k_film_snapshot(const float4* __restrict__ film, float4* __restrict__ snapshot, uint32_t pixels, uint32_t layer_mask, uint32_t own_first,
  uint32_t own_stride, uint32_t film_w, uint32_t film_h) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += gridDim.x * blockDim.x) {
    bool keep_count = true;
    if (own_stride > 1u) {
      const uint32_t fx = i % film_w, fy = i / film_w;
      const uint32_t pixel_id = fx + (film_h - 1u - fy) * film_w;  // film_index, dev_vcm.h
      keep_count = (pixel_id % own_stride) == own_first;
    }
#pragma unroll
    for (uint32_t layer = 0; layer < 4u; ++layer) {
      if ((layer_mask >> layer) & 1u) {
        float4 v = film[size_t(layer) * pixels + i];
        if ((layer == 0u) && (keep_count == false))
          v.w = 0.0f;
        snapshot[size_t(layer) * pixels + i] = v;
      }
    }
  }
}

__global__ void k_set_words(unsigned long long* dst, unsigned long long a, unsigned long long b) {
  if ((blockIdx.x == 0) && (threadIdx.x == 0))
    dst[0] = a, dst[1] = b;
}

void launch_set_words(hipStream_t stream, unsigned long long* dst, unsigned long long a, unsigned long long b) {
  hipLaunchKernelGGL(k_set_words, dim3(1), dim3(64), 0, stream, dst, a, b);
}

void launch_film_snapshot(hipStream_t stream, const float4* film, float4* snapshot, uint32_t pixels, uint32_t layer_mask, uint32_t own_first, uint32_t own_stride, uint32_t film_w, uint32_t film_h) {
  hipLaunchKernelGGL(k_film_snapshot, dim3(grid_for(pixels)), dim3(kBlockSize), 0, stream, film, snapshot, pixels, layer_mask, own_first, own_stride, film_w, film_h);
}

// ---------------------------------------------------------------------------------------------------------------
// Film::layer, film.cxx:381-418: float3 sums -> float4 (alpha 1); Result = max(0, camera + light)
// `counts` (nullable): the camera sums, whose w holds the iterations committed to each pixel (k_vcm_commit / k_pt_commit).
// A read-back that runs while iterations are in flight normalises every pixel by its own count: the host's iteration
// counter and the device's commits are not updated at the same instant.
__global__ void k_film_resolve(const float4* __restrict__ camera_sum, const float4* __restrict__ light_sum, float4* __restrict__ out, uint32_t pixel_count, float scale, int layer,
  const float4* __restrict__ counts) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixel_count)
    return;
  float4 c = camera_sum[i], l = light_sum[i];
  if (counts != nullptr) {
    const float n = counts[i].w;
    scale = (n > 0.0f) ? 1.0f / n : 0.0f;
  }
  float4 r;
  if (layer == 0)
    r = make_float4(c.x * scale, c.y * scale, c.z * scale, 1.0f);
  else if (layer == 1)
    r = make_float4(l.x * scale, l.y * scale, l.z * scale, 1.0f);
  else if (layer == 3)  // Film::layer(Normals) = buf * 0.5 + 0.5 (film.cxx:411)
    r = make_float4(c.x * scale * 0.5f + 0.5f, c.y * scale * 0.5f + 0.5f, c.z * scale * 0.5f + 0.5f, 1.0f);
  else
    r = make_float4(fmaxf(0.0f, (c.x + l.x) * scale), fmaxf(0.0f, (c.y + l.y) * scale), fmaxf(0.0f, (c.z + l.z) * scale), 1.0f);
  out[i] = r;
}

void launch_film_resolve(hipStream_t stream, const float4* camera_sum, const float4* light_sum, float4* out, uint32_t pixel_count, float scale, int layer, const float4* counts) {
  hipLaunchKernelGGL(k_film_resolve, dim3((pixel_count + 255u) / 256u), dim3(256), 0, stream, camera_sum, light_sum, out, pixel_count, scale, layer, counts);
}

// ---------------------------------------------------------------------------------------------------------------
// known-answer kernels (include/etx_hip.h: etx_hip_kat)
__global__ void k_kat(int which, const float* __restrict__ in, uint32_t count, float* __restrict__ out, const uint2* __restrict__ bluenoise) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  switch (which) {
    case 0: {
      Sampler s;
      s.init(__float_as_uint(in[2 * i + 0]), __float_as_uint(in[2 * i + 1]));
      out[4 * i + 0] = __uint_as_float(s.seed);
      out[4 * i + 1] = s.next();
      out[4 * i + 2] = s.next();
      out[4 * i + 3] = s.next();
      break;
    }
    case 1: {
      f3 r = offset_ray(f3{in[6 * i], in[6 * i + 1], in[6 * i + 2]}, f3{in[6 * i + 3], in[6 * i + 4], in[6 * i + 5]});
      out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
      break;
    }
    case 2: {
      Basis b = orthonormal_basis(f3{in[3 * i], in[3 * i + 1], in[3 * i + 2]});
      out[6 * i] = b.u.x, out[6 * i + 1] = b.u.y, out[6 * i + 2] = b.u.z, out[6 * i + 3] = b.v.x, out[6 * i + 4] = b.v.y, out[6 * i + 5] = b.v.z;
      break;
    }
    case 3: {
      f3 r = sample_cosine_distribution(f2{in[5 * i], in[5 * i + 1]}, f3{in[5 * i + 2], in[5 * i + 3], in[5 * i + 4]}, 1.0f);
      out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
      break;
    }
    case 4: {
      uint32_t idx = grid_cell_index(__float_as_int(in[4 * i]), __float_as_int(in[4 * i + 1]), __float_as_int(in[4 * i + 2]), __float_as_uint(in[4 * i + 3]));
      out[i] = __uint_as_float(idx);
      break;
    }
    case 5: {
      f2 r = sample_disk(f2{in[2 * i], in[2 * i + 1]});
      out[2 * i] = r.x, out[2 * i + 1] = r.y;
      break;
    }
    case 6: {  // in: pixel x, pixel y, sample (u32 bits) -> the six blue-noise numbers of vcm_camera_step
      f2 a, b, c;
      bluenoise_samples(bluenoise, __float_as_uint(in[3 * i]), __float_as_uint(in[3 * i + 1]), __float_as_uint(in[3 * i + 2]), a, b, c);
      out[6 * i] = a.x, out[6 * i + 1] = a.y, out[6 * i + 2] = b.x, out[6 * i + 3] = b.y, out[6 * i + 4] = c.x, out[6 * i + 5] = c.y;
      break;
    }
    default:
      break;
  }
}

void launch_kat(hipStream_t stream, int which, const float* in, uint32_t count, float* out, const uint2* bluenoise) {
  if (count == 0)
    return;
  hipLaunchKernelGGL(k_kat, dim3((count + 255u) / 256u), dim3(256), 0, stream, which, in, count, out, bluenoise);
}

}  // namespace etxd
