// kernels_connect.hip - the two variable-length expansions of the VCM camera pass, flattened for the GPU:
//
//   vertex connections  vcm_connect_to_light_path (vcm_shared.hxx:765-803): camera vertex x every vertex of the light
//       path of the SAME pixel. k_expand_pairs turns (camera vertex, light path) into a dense list of
//       (camera vertex, light vertex) pairs (wave prefix sum + one atomic per wave), k_connect_pairs evaluates one
//       pair per lane (2 BSDF evaluations + 2 reverse pdfs, the visibility segment goes to the shadow queue), so lanes do equal work
//       although light path lengths are geometric-ish (max over a wave ~5x the mean).
//   photon merge        VCMSpatialGridData::gather (vcm_shared.hxx:886-924): camera vertex x 8 hash cells x photons.
//       k_merge gives each (camera vertex, cell) its own lane: 8x more independent photon streams in flight per
//       vertex, partial sums folded with three xor-shuffles.
// Both accumulate straight into the film sums (float atomics). Roofline: HBM/L2 latency + fp32 VALU, no MFMA.
#include "kernels.h"
#include "dev_vcm.h"

namespace etxd {

#define ETX_WAVE_LOOP(COUNT)                                                          \
  const uint32_t lane_ = threadIdx.x & 63u;                                            \
  const uint32_t stride_ = gridDim.x * blockDim.x;                                     \
  for (uint32_t base_ = blockIdx.x * blockDim.x + threadIdx.x - lane_; base_ < (COUNT); base_ += stride_)

static uint32_t grid_for(uint32_t capacity) {
  return min(kPersistentBlocks, (capacity + kBlockSize - 1) / kBlockSize);
}

ETX_DEV bool material_is_diffuse(const DScene& scene, uint32_t tri) {
  return material_is_lambert(scene.materials[__float_as_uint(scene.tri_shade[size_t(tri) * kTriShadeStride + 6u].w)]);
}

// ---------------------------------------------------------------------------------------------------------------
// (camera vertex, light path) -> pairs
// kVcmRecords: the camera vertex records are VCM's (pos_info.w holds kCv* flags). (The bidirectional integrator has an expansion of its own since round 6:
// k_bdpt_expand_pairs, kernels_bdpt.hip - two lists by BSDF class, chunked index lists instead of the `next` links.)
template <bool kVcmRecords>
__global__ __launch_bounds__(kBlockSize) void k_expand_pairs(Pipeline p, VcmParams it) {
  __shared__ uint32_t s_wave_total[kBlockSize / 64u];
  __shared__ uint32_t s_base;
  const uint32_t count = min(p.counters[kCntCameraVertices], p.cv_capacity);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6u;
  ETX_BLOCK_LOOP(count, i) {
    uint32_t head = kInvalid, k = 0, path = 0;
    uint4 first = make_uint4(kInvalid, 0u, kInvalid, kInvalid);  // head, length and the path's first two vertices: one 16-byte load of its table row
    if (i < count) {
      path = __float_as_uint(p.cv.mis_pixel[i].w);
      const bool skip = kVcmRecords && (__float_as_uint(p.cv.pos_info[i].w) & kCvNoConnect);  // merge-only record of a Christensen-Burley vertex
      if (skip == false)
        first = p.light_path_table[size_t(path) * (p.path_table_entries >> 2u)];
      head = first.x;
      k = first.y;
    }
    // workgroup exclusive prefix sum of k: wave scan, then one reservation for all four waves
    uint32_t incl = k;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(incl, d);
      if (lane >= d)
        incl += t;
    }
    if (lane == 63u)
      s_wave_total[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0u) {
      uint32_t total = 0;
#pragma unroll
      for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
        total += s_wave_total[w];
      s_base = total ? atomicAdd(p.counters + kCntPairs, total) : 0u;
      if (total)
        atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatPairs), (unsigned long long)total);
    }
    __syncthreads();
    uint32_t base = s_base + incl - k;
#pragma unroll
    for (uint32_t w = 0; w < kBlockSize / 64u; ++w)
      base += (w < wave) ? s_wave_total[w] : 0u;
    __syncthreads();
    if (base + k > p.pair_capacity) {
      if (k)
        atomicOr(p.counters + kCntOverflow, kOverflowPairs);
      continue;
    }
    if (k == 0u)
      continue;
    // the first path_table_entries vertices come from the path table (independent 16-byte loads), only longer
    // paths walk the list from the head down to that index
    const uint32_t table_entries = p.path_table_entries - kPathRowHeader;  // vertices a row holds behind its two header words
    const uint4* table = p.light_path_table + size_t(path) * (p.path_table_entries >> 2u);
    const uint32_t from_table = min(k, table_entries);
    p.pairs[base] = make_uint2(i, first.z);  // k >= 1 here
    if (from_table > 1u)
      p.pairs[base + 1u] = make_uint2(i, first.w);
    for (uint32_t q = 1; (q << 2u) < from_table + kPathRowHeader; ++q) {  // word 4 q + c of the row = vertex 4 q + c - 2 of the path
      const uint4 t = table[q];
      const uint32_t j = (q << 2u) - kPathRowHeader;
      p.pairs[base + j] = make_uint2(i, t.x);
      if (j + 1u < from_table)
        p.pairs[base + j + 1u] = make_uint2(i, t.y);
      if (j + 2u < from_table)
        p.pairs[base + j + 2u] = make_uint2(i, t.z);
      if (j + 3u < from_table)
        p.pairs[base + j + 3u] = make_uint2(i, t.w);
    }
    uint32_t vi = head;
    for (uint32_t j = k; j > table_entries; --j) {
      p.pairs[base + j - 1u] = make_uint2(i, vi);
      vi = p.lv.next(vi);
    }
  }
}

// `classify`: the scene holds connectible materials other than Lambert, so a pair is routed by the classes of its two vertices (two dependent
// gathers each). In a scene of Lambert / delta materials every stored vertex is Lambert or a medium vertex: nothing to look up.
template <bool kDiffuseOnly>
__global__ __launch_bounds__(kBlockSize) void k_connect_pairs(Pipeline p, VcmParams it, uint32_t classify) {
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  const uint32_t count = pair_list_count(p);
  ETX_BLOCK_LOOP(count, i) {
    ShadowRequest request;
    bool queue = false;
    if (i < count) {
      const uint2 pair = p.pairs[i];
      LightVertex lv = load_light_vertex(p.lv, pair.y);
      const uint32_t cam_tri = __float_as_uint(p.cv.hit[pair.x].w);
      const bool cam_exit = (__float_as_uint(p.cv.thr_depth[pair.x].w) & kCvExitMaterialBit) != 0u;  // subsurface exit point: white Lambert
      bool all_diffuse = (classify == 0u) || (((cam_tri == kInvalid) || cam_exit || material_is_diffuse(scene, cam_tri)) && (lv.is_medium() || material_is_diffuse(scene, lv.tri)));
      if (all_diffuse == kDiffuseOnly) {
        CameraVertex cv = load_camera_vertex(p, scene, pair.x);
        const uint32_t target_path_length = cv.st.depth + lv.index_in_path + 2u;  // vcm_shared.hxx:774
        if ((target_path_length >= scene.min_path_length) && (target_path_length <= scene.max_path_length)) {
          // the reference evaluates every connection of a vertex with the path's sampler; decorrelate per pair - by the light vertex's index in
          // ITS path (the pairs of a camera vertex are the vertices of one light path), not by its pool slot, which differs from run to run
          cv.st.sampler.seed = Sampler::random_seed(cv.st.sampler.seed, lv.index_in_path);
          f3 target_position, value;
          if (vcm_connect_to_light_vertex<kDiffuseOnly>(scene, cv.st, lv, it, cv.at_medium, &cv.isect, cv.medium_pos, cv.st.sampler, target_position, value)) {
            f3 p0 = cv.medium_pos;
            if (cv.at_medium == false)
              p0 = shading_pos(scene, scene.triangles[cv.isect.tri], cv.isect.bc, normalize(target_position - cv.isect.pos));
            request = {p0, cv.at_medium ? lv.pos : target_position, value * spectral_film_weight(scene, cv.st.wavelength), cv.st.medium, film_index(it, cv.st.id), cv.st.wavelength};
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

// ---------------------------------------------------------------------------------------------------------------
// Endpoint connections of the general / subsurface shading groups (pipeline.h EndpointQueue): one request per lane,
// vcm_connect_to_camera (light pass) or vcm_connect_to_light (camera pass) over every BSDF class, the visibility
// segment goes to the shadow queue like every other connection.
template <bool kCameraPass>
__global__ __launch_bounds__(kBlockSize) void k_connect_endpoints(Pipeline p, VcmParams it) {
  __shared__ BlockScratch s_scratch;
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntEndpoints], p.endpoints.capacity);
  if ((blockIdx.x == 0) && (threadIdx.x == 0) && (count != 0u))
    atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + kStatEndpoints), (unsigned long long)count);
  ETX_BLOCK_LOOP(count, i) {
    ShadowRequest request;
    bool queue = false;
    if (i < count) {
      CameraVertex v = load_endpoint(p, scene, i);
      if (kCameraPass)
        queue = vcm_connect_to_light<false>(scene, it, v.at_medium, &v.isect, v.medium_pos, v.st, film_index(it, v.st.id), request, scene.materials[v.at_medium ? 0u : v.isect.material]);
      else
        queue = vcm_connect_to_camera<false>(scene, it, v.at_medium, &v.isect, v.medium_pos, v.st, request, scene.materials[v.at_medium ? 0u : v.isect.material]);
    }
    const uint32_t slot = block_compact_slot(queue, p.counters + kCntShadow, s_scratch);
    if (queue)
      write_shadow(p, slot, request);
  }
}

void launch_connect_endpoints(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool camera_pass, uint32_t max_items) {
  const uint32_t blocks = max(1u, grid_for(min(max_items, p.endpoints.capacity)));
  if (camera_pass)
    hipLaunchKernelGGL(k_connect_endpoints<true>, dim3(blocks), dim3(kBlockSize), 0, stream, p, it);
  else
    hipLaunchKernelGGL(k_connect_endpoints<false>, dim3(blocks), dim3(kBlockSize), 0, stream, p, it);
}

void launch_connect(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool generic_materials, uint32_t max_items) {
  max_items = min(max_items, p.capacity);
  const uint32_t blocks = max(1u, grid_for(max_items));
  const uint32_t pair_blocks = max(1u, grid_for(uint32_t(min(uint64_t(max_items) * 8ull, uint64_t(p.pair_capacity)))));
  hipLaunchKernelGGL(k_expand_pairs<true>, dim3(blocks), dim3(kBlockSize), 0, stream, p, it);
  hipLaunchKernelGGL(k_connect_pairs<true>, dim3(pair_blocks), dim3(kBlockSize), 0, stream, p, it, generic_materials ? 1u : 0u);
  if (generic_materials)
    hipLaunchKernelGGL(k_connect_pairs<false>, dim3(pair_blocks), dim3(kBlockSize), 0, stream, p, it, 1u);
}

// ---------------------------------------------------------------------------------------------------------------
// VCMSpatialGridData::gather / gather_index, vcm_shared.hxx:829-924 : one lane per (camera vertex, hash cell)
ETX_DEV bool merge_cell_range(const Pipeline& p, const GridParams& g, const f3& pos, uint32_t c, uint32_t& range_begin, uint32_t& range_end) {
  if ((pos.x < g.bbox_min.x) || (pos.y < g.bbox_min.y) || (pos.z < g.bbox_min.z) || (pos.x > g.bbox_max.x) || (pos.y > g.bbox_max.y) || (pos.z > g.bbox_max.z))
    return false;  // BoundingBox::contains, vcm_shared.hxx:891-893
  f3 m = (pos - g.bbox_min) / g.cell_size;
  f3 mf = {floorf(m.x), floorf(m.y), floorf(m.z)};
  f3 md = m - mf;
  int32_t cx = int32_t(mf.x) + ((c & 1u) ? ((md.x < 0.5f) ? -1 : +1) : 0);
  int32_t cy = int32_t(mf.y) + ((c & 2u) ? ((md.y < 0.5f) ? -1 : +1) : 0);
  int32_t cz = int32_t(mf.z) + ((c & 4u) ? ((md.z < 0.5f) ? -1 : +1) : 0);
  // two neighbour offsets can hash to the same cell: the reference then visits that cell twice, so does this kernel
  const uint32_t cell = grid_cell_index(cx, cy, cz, g.hash_mask);
  range_begin = (cell == 0u) ? 0u : p.grid.cell_ends[cell - 1u];
  range_end = p.grid.cell_ends[cell];
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Cell-ordered merge. After the first bounce camera vertices are spatially random, so every (vertex, cell) lane pulled
// its photons from HBM (PMC: 42 GB per iteration, 22 % L2 hits). The vertices of a bounce are therefore counting-sorted
// by a coarse spatial bucket (64^3 Morton-ordered blocks) and the merge walks them in that order, each XCD owning one
// contiguous eighth of the list so that its private L2 sees a compact working set (PMC after: 13 GB, 59 % L2 hits).
// Only indices are sorted (4 B per vertex). The histogram is taken where the vertices are written (store_camera_vertex, dev_vcm_steps.h:
// one atomic per vertex in the shade kernel instead of a pass over the pool) and zeroed again by k_merge_diffuse, which no longer needs it:
// per bounce the sort costs two scan launches and one scatter (merge_bucket / merge_candidate: dev_vcm.h).
// Exclusive scan of the 2^18 bucket counters in two small launches (one block over 1 MiB took 96 us per bounce):
// kMergeScanBlocks workgroups of 256 threads, one uint4 per thread; the first launch leaves every workgroup's total
// behind the counters, the second scans those totals in LDS (every workgroup redundantly) and then its own 1024 counters.
constexpr uint32_t kMergeScanBlocks = kMergeBuckets / (4u * kBlockSize);
static_assert(kMergeScanBlocks <= kBlockSize, "group totals are scanned by one workgroup-wide pass");

ETX_DEV uint32_t block_exclusive_scan(uint32_t value, uint32_t* s_wave_sums, uint32_t& block_total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6u;
  uint32_t incl = value;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d);
    if (lane >= d)
      incl += t;
  }
  if (lane == 63u)
    s_wave_sums[wave] = incl;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kBlockSize / 64u; ++w) {
    const uint32_t ws = s_wave_sums[w];
    before += (w < wave) ? ws : 0u;
    total += ws;
  }
  __syncthreads();
  block_total = total;
  return before + incl - value;
}

__global__ __launch_bounds__(kBlockSize) void k_merge_scan_totals(Pipeline p) {
  __shared__ uint32_t s_wave_sums[kBlockSize / 64u];
  const uint4 v = reinterpret_cast<const uint4*>(p.merge_buckets)[blockIdx.x * kBlockSize + threadIdx.x];
  uint32_t total = 0;
  block_exclusive_scan(v.x + v.y + v.z + v.w, s_wave_sums, total);
  if (threadIdx.x == 0)
    p.merge_buckets[kMergeBuckets + 1u + blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlockSize) void k_merge_scan(Pipeline p) {
  __shared__ uint32_t s_wave_sums[kBlockSize / 64u];
  __shared__ uint32_t s_group_offset;
  const uint32_t group_total = (threadIdx.x < kMergeScanBlocks) ? p.merge_buckets[kMergeBuckets + 1u + threadIdx.x] : 0u;
  uint32_t all = 0;
  const uint32_t group_before = block_exclusive_scan(group_total, s_wave_sums, all);
  if (threadIdx.x == blockIdx.x)
    s_group_offset = group_before;
  uint4* mine = reinterpret_cast<uint4*>(p.merge_buckets) + blockIdx.x * kBlockSize + threadIdx.x;
  const uint4 v = *mine;
  uint32_t unused = 0;
  uint32_t running = block_exclusive_scan(v.x + v.y + v.z + v.w, s_wave_sums, unused);  // contains the barrier for s_group_offset
  running += s_group_offset;
  uint4 o;
  o.x = running, running += v.x;
  o.y = running, running += v.y;
  o.z = running, running += v.z;
  o.w = running;
  *mine = o;
  if ((blockIdx.x == 0) && (threadIdx.x == 0)) {
    p.counters[kCntMergeVertices] = all;
    p.counters[kCntPairs] = 0u;  // the pair list is the merge's from here on (k_connect_pairs of this bounce is done): k_merge_filter_generic appends to it
  }
}

__global__ __launch_bounds__(kBlockSize) void k_merge_scatter(Pipeline p) {
  const GridParams g = *p.grid_params;
  if ((g.valid == 0u) || (g.photon_count == 0u))
    return;
  const uint32_t count = min(p.counters[kCntCameraVertices], p.cv_capacity);
  const uint32_t max_path_length = p.scene.max_path_length;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float4 pi = p.cv.pos_info[i];
    const f3 pos = {pi.x, pi.y, pi.z};
    if (merge_candidate(g, max_path_length, __float_as_uint(pi.w), pos))
      p.merge_order[atomicAdd(p.merge_buckets + merge_bucket(g, pos), 1u)] = i;
  }
}

// XCD-aware item range: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only), so XCD x walks
// the x-th eighth of the sorted item list. (Handing out chunks dynamically from one cursor was measured 40 % slower.)
#define ETX_XCD_RANGE_LOOP(ITEMS)                                                                    \
  const uint32_t lane_ = threadIdx.x & 63u;                                                          \
  const uint32_t xcd_ = blockIdx.x & 7u, local_block_ = blockIdx.x >> 3u, blocks_per_xcd_ = gridDim.x >> 3u; \
  const uint32_t seg_ = (((ITEMS) + 7u) / 8u + 63u) & ~63u;                                          \
  const uint32_t seg_begin_ = xcd_ * seg_, seg_end_ = min((ITEMS), seg_begin_ + seg_);              \
  for (uint32_t base_ = seg_begin_ + local_block_ * blockDim.x + threadIdx.x - lane_; base_ < seg_end_; base_ += blocks_per_xcd_ * blockDim.x)

// Diffuse camera vertices (the common case): everything the loop needs comes from five float4 of the vertex record.
//
// Work distribution: a wave takes 8 vertices x 8 cells = 64 photon ranges. Range lengths vary wildly (0..100+), so
// instead of one lane looping over "its" range (PMC: ~16 % of the lanes active per load instruction) the 64 ranges are
// flattened: an exclusive wave scan of the lengths gives every photon of the batch a global element index, lane l
// processes elements l, l+64, ... and finds its range by a 6-step binary search over the scanned offsets
// (ds_bpermute), reads that range's vertex record from LDS, and adds accepted contributions to the vertex'
// accumulator with LDS float atomics. Every load instruction then serves 64 photons.
struct MergeSlot {   // per range, in LDS
  float4 pos_depth;  // vertex position, total_path_depth bits
  float4 nrm_dvm;    // shading normal, d_vm
  float4 wi_wcam;    // w_i, d_vcm * vc_weight
  float4 fthr;       // albedo/pi * throughput, unused
};

__global__ __launch_bounds__(kBlockSize) void k_merge_diffuse(Pipeline p, VcmParams it) {
  __shared__ MergeSlot s_slot[kBlockSize];
  __shared__ float s_acc[kBlockSize / 64][8][4];
  __shared__ uint2 s_ring[kBlockSize / 64][128];  // (photon, distance^2 bits)
  __shared__ uint32_t s_ring_range[kBlockSize / 64][128];
  const DScene& scene = p.scene;
  const uint32_t count = min(p.counters[kCntMergeVertices], p.cv_capacity);
  const GridParams g = *p.grid_params;
  // the bucket cursors k_merge_scatter left behind: zero for the next bounce's histogram (nothing below reads them)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= kMergeBuckets; i += gridDim.x * blockDim.x)
    p.merge_buckets[i] = 0u;
  if ((g.valid == 0u) || (g.photon_count == 0u))
    return;
  const uint32_t items = count * 8u;
  const bool use_mis = opt_enable_mis(it);
  const bool use_epan = it.kernel == ETX_VCM_KERNEL_EPANECHNIKOV;
  const uint32_t max_path_length = scene.max_path_length;
  const uint32_t wave = threadIdx.x >> 6u;
  MergeSlot* slots = s_slot + wave * 64u;
  uint2* ring = s_ring[wave];
  uint32_t* ring_range = s_ring_range[wave];
  unsigned long long examined = 0, merged_count = 0;
  ETX_XCD_RANGE_LOOP(items) {
    const uint32_t item = base_ + lane_;
    const uint32_t c = item & 7u;
    uint32_t range_begin = 0, range_len = 0, pixel = 0;
    float wavelength = 0.0f;
    if (item < seg_end_) {
      const uint32_t vertex = p.merge_order[item >> 3u];
      const float4 pi = p.cv.pos_info[vertex];
      const uint32_t info = __float_as_uint(pi.w);
      const uint32_t depth = info >> 8u;
      const f3 pos = {pi.x, pi.y, pi.z};
      uint32_t range_end = 0;
      if ((info & kCvDiffuse) && (depth + 1u <= max_path_length) && merge_cell_range(p, g, pos, c, range_begin, range_end) && (range_begin < range_end)) {
        range_len = range_end - range_begin;
        const float4 nv = p.cv.nrm_dvm[vertex];
        const float4 wv = p.cv.wi_medium[vertex];
        const float4 fv = p.cv.fthr_dvcm[vertex];
        slots[lane_].pos_depth = make_float4(pi.x, pi.y, pi.z, __uint_as_float(depth));
        slots[lane_].nrm_dvm = nv;
        slots[lane_].wi_wcam = make_float4(wv.x, wv.y, wv.z, fv.w * it.vc_weight);
        // spectral mode: c_value = (func x throughput / sampling_pdf).to_rgb() (vcm_shared.hxx:869), channel-wise product with the photon's RGB
        const f3 cw = f3{fv.x, fv.y, fv.z} * spectral_film_weight(scene, p.cv.wavelength[vertex]);
        slots[lane_].fthr = make_float4(cw.x, cw.y, cw.z, fv.w);
      }
      if (c == 0u)
        pixel = __float_as_uint(p.cv.mis_pixel[vertex].w), wavelength = p.cv.wavelength[vertex];
    }
    if (lane_ < 32u)
      (&s_acc[wave][0][0])[lane_] = 0.0f;
    // exclusive scan of the range lengths
    uint32_t incl = range_len;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(incl, d);
      if (lane_ >= d)
        incl += t;
    }
    const uint32_t total = __shfl(incl, 63);
    const uint32_t offset = incl - range_len;
    __threadfence_block();  // slots / accumulators written before any lane of this wave reads them
    if (lane_ == 0)
      examined += total;
    // Two phases per batch. The distance / path-length filter accepts about one photon in five, so running the BSDF and
    // MIS arithmetic right behind it would leave four lanes in five idle: accepted (range, photon) pairs go to a
    // 128-entry LDS ring instead and are evaluated 64 at a time with every lane busy.
    uint32_t ring_head = 0, ring_tail = 0;  // wave-uniform
    auto evaluate = [&](uint32_t entries) {
      if (lane_ < entries) {
        const uint2 en = ring[(ring_head + lane_) & 127u];
        const uint32_t r = ring_range[(ring_head + lane_) & 127u], j = en.x;
        const float distance_squared = __uint_as_float(en.y);
        const MergeSlot& sl = slots[r];
        const float4 nv = sl.nrm_dvm;
        const f3 nrm = {nv.x, nv.y, nv.z};
        const float4 nd = p.grid.nrm_dvcm(j);
        const float4 wv = sl.wi_wcam;
        const f3 w_i = {wv.x, wv.y, wv.z};
        const float4 wd = p.grid.win_dvm(j);
        const f3 wi = {wd.x, wd.y, wd.z};
        const f3 n_front = dot(nrm, w_i) < 0.0f ? nrm : -nrm;  // get_normal_frame, bsdf.hxx:37-40
        const float cos_o = -dot(n_front, wi);                 // DiffuseBSDF::evaluate(-wi), bsdf_various.hxx:97-106
        if ((dot(nrm, f3{nd.x, nd.y, nd.z}) > kEpsilon) && (cos_o > kEpsilon)) {
          const float pdf = kInvPi * cos_o;
          // reverse_pdf: roles swapped, w_i' = wi, w_o' = -w_i (scene_bsdf.hxx:82-92 + bsdf_various.hxx:113-120)
          const f3 n_rev = dot(nrm, wi) < 0.0f ? nrm : -nrm;
          const float n_dot_o = -dot(n_rev, w_i);
          const float rev_pdf = (n_dot_o <= kEpsilon) ? 0.0f : kInvPi * n_dot_o;
          const float w_light = nd.w * it.vc_weight + wd.w * pdf;
          const float w_camera = wv.w + nv.w * rev_pdf;
          const float weight = use_mis ? (1.0f / (1.0f + w_light + w_camera)) : 1.0f;
          const float kernel_weight = use_epan ? fmaxf(2.0f * (1.0f - distance_squared * g.inv_radius_squared), 0.0f) : 1.0f;
          const float4 lt = p.grid.thr(j);
          const float4 fv = sl.fthr;
          const float k = kernel_weight * weight;
          float* acc = s_acc[wave][r >> 3u];
          atomicAdd(acc + 0, fv.x * lt.x * k);
          atomicAdd(acc + 1, fv.y * lt.y * k);
          atomicAdd(acc + 2, fv.z * lt.z * k);
          merged_count++;
        }
      }
      ring_head += entries;
    };
    for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
      const uint32_t e = e0 + lane_;
      // binary search: last range r with offset[r] <= e
      uint32_t r = 0;
#pragma unroll
      for (uint32_t step = 32u; step > 0u; step >>= 1u) {
        const uint32_t cand = r + step;
        const uint32_t o = __shfl(offset, cand & 63u);
        if (o <= e)
          r = cand;
      }
      const uint32_t r_begin = __shfl(range_begin, r);
      const uint32_t r_offset = __shfl(offset, r);
      bool accept = false;
      uint32_t j = 0;
      float distance_squared = 0.0f;
      if (e < total) {
        j = r_begin + (e - r_offset);
        const float4 pl = p.grid.pos_len[j];
        const float4 sp = slots[r].pos_depth;
        const f3 d = f3{pl.x, pl.y, pl.z} - f3{sp.x, sp.y, sp.z};
        distance_squared = dot(d, d);
        accept = (distance_squared <= g.radius_squared) && (__float_as_uint(pl.w) + __float_as_uint(sp.w) + 1u <= max_path_length);
      }
      const unsigned long long mask = __ballot(accept);
      if (accept) {
        const uint32_t at = (ring_tail + __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32u), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u))) & 127u;
        ring[at] = make_uint2(j, __float_as_uint(distance_squared));
        ring_range[at] = r;
      }
      ring_tail += uint32_t(__popcll(mask));
      __threadfence_block();
      if (ring_tail - ring_head >= 64u)
        evaluate(64u);
    }
    if (ring_tail != ring_head)
      evaluate(ring_tail - ring_head);
    __threadfence_block();
    if (((lane_ & 7u) == 0u) && (item < seg_end_)) {
      const float* acc = s_acc[wave][lane_ >> 3u];
      const f3 merged = {acc[0], acc[1], acc[2]};
      if ((merged.x != 0.0f) || (merged.y != 0.0f) || (merged.z != 0.0f))
        film_add(p, p.camera_sum + film_index(it, pixel), merged * it.vm_normalization);
    }
    __threadfence_block();  // accumulators are zeroed again at the top of the next batch
  }
  __shared__ unsigned long long s_stat;
  block_stat_add(p, kBlockStatExamined, examined, &s_stat);
  block_stat_add(p, kBlockStatMerged, merged_count, &s_stat);
}

// Every other connectible material: generic BSDF evaluation per accepted photon (stochastic for the Heitz models: a random walk of up to 16 steps).
// Round 5 did filter and evaluation in ONE kernel shaped like k_merge_diffuse; on the configs[2] family it was 41.5 % of the step at 3 % occupancy and
// 7 % VALU busy (profiles/round5_pmc_gems_1lane_summary.txt): the vertex list is sorted in space, the gems sit in one corner of it, so a handful of
// workgroups evaluated every accepted photon while the rest of the grid found nothing to do. Now two kernels, like the connections
// (k_expand_pairs -> k_connect_pairs):
//   k_merge_filter_generic  the distance / path-length filter of VCMSpatialGridData::gather (vcm_shared.hxx:829-851) over the flattened ranges of the
//       generic vertices - 16 bytes per photon examined, no BSDF code, 40 VGPRs; accepted (camera vertex, photon) pairs are appended to the PAIR LIST
//       (free at this point of a bounce: k_connect_pairs has consumed it; k_merge_scan zeroes its counter), 64 at a time through the LDS ring with one
//       reservation per 64 pairs;
//   k_merge_eval_generic    one pair per lane over the dense list, whichever workgroup: bsdf::evaluate + reverse pdf + MIS (:852-884), contributions
//       of a vertex folded across the wavefront (its pairs are neighbours in the list) before the film atomics.
__global__ __launch_bounds__(kBlockSize) void k_merge_filter_generic(Pipeline p, VcmParams it) {
  __shared__ float4 s_pos_depth[kBlockSize / 64][8];  // per camera vertex of the batch: position, total path depth bits
  __shared__ uint32_t s_vertex[kBlockSize / 64][8];
  __shared__ uint2 s_ring[kBlockSize / 64][128];      // (camera vertex, photon)
  __shared__ unsigned long long s_stat;
  const uint32_t count = min(p.counters[kCntMergeVertices], p.cv_capacity);
  const GridParams g = *p.grid_params;
  if ((g.valid == 0u) || (g.photon_count == 0u))
    return;
  const uint32_t items = count * 8u;
  const uint32_t max_path_length = p.scene.max_path_length;
  const uint32_t wave = threadIdx.x >> 6u;
  float4* pos_depth = s_pos_depth[wave];
  uint32_t* vertex_of = s_vertex[wave];
  uint2* ring = s_ring[wave];
  unsigned long long examined = 0;
  ETX_XCD_RANGE_LOOP(items) {
    const uint32_t item = base_ + lane_;
    const uint32_t c = item & 7u;
    uint32_t range_begin = 0, range_len = 0;
    if (item < seg_end_) {
      const uint32_t vertex = p.merge_order[item >> 3u];
      const float4 pi = p.cv.pos_info[vertex];
      const uint32_t info = __float_as_uint(pi.w);
      const uint32_t depth = info >> 8u;
      const bool generic = ((info & (kCvDiffuse | kCvMedium)) == 0u) && (depth + 1u <= max_path_length);
      uint32_t range_end = 0;
      if (generic && merge_cell_range(p, g, f3{pi.x, pi.y, pi.z}, c, range_begin, range_end) && (range_begin < range_end))
        range_len = range_end - range_begin;
      if (generic && (c == 0u)) {
        pos_depth[lane_ >> 3u] = make_float4(pi.x, pi.y, pi.z, __uint_as_float(depth));
        vertex_of[lane_ >> 3u] = vertex;
      }
    }
    if (__ballot(range_len != 0u) == 0ull)
      continue;  // no generic vertex with photons in this batch (the common case: the list is sorted in space)
    uint32_t incl = range_len;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(incl, d);
      if (lane_ >= d)
        incl += t;
    }
    const uint32_t total = __shfl(incl, 63);
    const uint32_t offset = incl - range_len;
    __threadfence_block();
    if (lane_ == 0)
      examined += total;
    uint32_t ring_head = 0, ring_tail = 0;  // wave-uniform
    auto flush = [&](uint32_t entries) {
      uint32_t base = 0;
      if (lane_ == 0)
        base = atomicAdd(p.counters + kCntPairs, entries);
      base = __shfl(base, 0);
      if (base + entries > p.pair_capacity) {
        if (lane_ == 0)
          atomicOr(p.counters + kCntOverflow, kOverflowPairs);  // the iteration is discarded, the pools grow, it is rendered again (host_api.cpp execute_iteration)
      } else if (lane_ < entries) {
        p.pairs[base + lane_] = ring[(ring_head + lane_) & 127u];
      }
      ring_head += entries;
    };
    for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
      const uint32_t e = e0 + lane_;
      uint32_t r = 0;
#pragma unroll
      for (uint32_t step = 32u; step > 0u; step >>= 1u) {
        const uint32_t cand = r + step;
        const uint32_t o = __shfl(offset, cand & 63u);
        if (o <= e)
          r = cand;
      }
      const uint32_t r_begin = __shfl(range_begin, r);
      const uint32_t r_offset = __shfl(offset, r);
      bool accept = false;
      uint32_t j = 0;
      if (e < total) {
        j = r_begin + (e - r_offset);
        const float4 pl = p.grid.pos_len[j];
        const float4 sp = pos_depth[r >> 3u];
        const f3 d = f3{pl.x, pl.y, pl.z} - f3{sp.x, sp.y, sp.z};
        accept = (dot(d, d) <= g.radius_squared) && (__float_as_uint(pl.w) + __float_as_uint(sp.w) + 1u <= max_path_length);
      }
      const unsigned long long mask = __ballot(accept);
      if (accept) {
        const uint32_t at = (ring_tail + __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32u), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u))) & 127u;
        ring[at] = make_uint2(vertex_of[r >> 3u], j);
      }
      ring_tail += uint32_t(__popcll(mask));
      __threadfence_block();
      if (ring_tail - ring_head >= 64u)
        flush(64u);
    }
    if (ring_tail != ring_head)
      flush(ring_tail - ring_head);
    __threadfence_block();  // the vertex slots are rewritten at the top of the next batch
  }
  block_stat_add(p, kBlockStatExamined, examined, &s_stat);
}

struct MergeVertex {  // what the evaluation reads of a camera vertex, per lane in LDS: the rebuilt intersection would otherwise sit in ~40 VGPRs across the BSDF code
  f3 pos, nrm, tan, btn, w_i, thr_film;
  f2 tex;
  float wavelength, w_camera_base, d_vm;
  uint32_t medium, material, seed;
};

__global__ __launch_bounds__(kBlockSize) void k_merge_eval_generic(Pipeline p, VcmParams it) {
  __shared__ MergeVertex s_vertex[kBlockSize];
  __shared__ unsigned long long s_stat;
  const DScene& scene = p.scene;
  const GridParams g = *p.grid_params;
  const uint32_t count = ((g.valid == 0u) || (g.photon_count == 0u)) ? 0u : pair_list_count(p);
  const bool use_mis = opt_enable_mis(it);
  const bool use_epan = it.kernel == ETX_VCM_KERNEL_EPANECHNIKOV;
  unsigned long long merged_count = 0;
  MergeVertex& v = s_vertex[threadIdx.x];
  ETX_WAVE_LOOP(count) {
    const uint32_t i = base_ + lane_;
    uint32_t vertex = kInvalid, pixel = 0, j = 0;
    f3 value = mk3(0.0f);
    if (i < count) {
      const uint2 pair = p.pairs[i];
      vertex = pair.x, j = pair.y;
      const CameraVertex cv = load_camera_vertex(p, scene, vertex);
      pixel = cv.st.id;
      const float4 pi = p.cv.pos_info[vertex];  // the position the filter measured from
      v.pos = {pi.x, pi.y, pi.z}, v.nrm = cv.isect.nrm, v.tan = cv.isect.tan, v.btn = cv.isect.btn, v.w_i = cv.isect.w_i, v.tex = cv.isect.tex;
      // c_value = (func x throughput / sampling_pdf).to_rgb(), vcm_shared.hxx:869
      v.thr_film = cv.st.throughput * spectral_film_weight(scene, cv.st.wavelength);
      v.wavelength = cv.st.wavelength;
      v.w_camera_base = cv.st.d_vcm * it.vc_weight;
      v.d_vm = cv.st.d_vm;
      v.medium = cv.st.medium, v.material = cv.isect.material, v.seed = cv.st.sampler.seed;
    }
    __threadfence_block();  // the record is read back from LDS where it is used, not carried in registers
    if (i < count) {
      const float4 nd = p.grid.nrm_dvcm(j);
      if (dot(v.nrm, f3{nd.x, nd.y, nd.z}) > kEpsilon) {
        const float4 wd = p.grid.win_dvm(j);
        const f3 wi = {wd.x, wd.y, wd.z};
        const etx_abi_material& mat = scene.materials[v.material];
        const BsdfData camera_data = {v.nrm, v.tan, v.btn, v.tex, v.w_i, v.medium, kPathCamera, v.wavelength};
        Sampler smp;  // the reference continues the path's stream through all photons; here one stream per (vertex, photon), keyed by the photon's
        smp.seed = Sampler::random_seed(v.seed, __float_as_uint(wd.x) ^ (__float_as_uint(wd.w) * 0x9e3779b9u));  // own values, not by where the sort put it
        smp.fixed_u = smp.fixed_v = smp.fixed_w = 0.0f;
        const BsdfEval camera_bsdf = bsdf_evaluate_general(scene, camera_data, -wi, v.material, smp);
        if (camera_bsdf.valid()) {
          const float rev_pdf = bsdf_reverse_pdf_s<false>(scene, camera_data, -wi, mat, smp);
          const float w_light = nd.w * it.vc_weight + wd.w * camera_bsdf.pdf;
          const float w_camera = v.w_camera_base + v.d_vm * rev_pdf;
          const float weight = use_mis ? (1.0f / (1.0f + w_light + w_camera)) : 1.0f;
          const float4 pl = p.grid.pos_len[j];
          const f3 d = f3{pl.x, pl.y, pl.z} - v.pos;
          const float kernel_weight = use_epan ? fmaxf(2.0f * (1.0f - dot(d, d) * g.inv_radius_squared), 0.0f) : 1.0f;
          const float4 lt = p.grid.thr(j);
          value = camera_bsdf.func * v.thr_film * f3{lt.x, lt.y, lt.z} * (kernel_weight * weight);
          merged_count++;
        }
      }
    }
    // the pairs of a vertex are neighbours in the list (one flush of the filter's ring holds a vertex' photons back to back): fold each RUN of equal
    // vertices into its first lane, one set of film atomics per run. Runs, not vertices: two flushes of one vertex may sit in one wavefront with another
    // vertex' pairs between them (the first version compared vertex indices across the distance d and counted such a second run twice - once folded into
    // the first run's head across the gap, once by its own head: one pixel in 25 000 off by one contribution, tests/test_gpu_pixel_sharding.py caught it)
    const uint32_t before = __shfl_up(vertex, 1);
    const bool head = (lane_ == 0u) || (before != vertex);
    const unsigned long long heads = __ballot(head);
    const uint32_t run = uint32_t(__popcll(heads & (~0ull >> (63u - lane_))));  // heads at or before this lane: equal only within one contiguous run
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t other = __shfl_down(run, d);
      const float ox = __shfl_down(value.x, d), oy = __shfl_down(value.y, d), oz = __shfl_down(value.z, d);
      if ((lane_ + d < 64u) && (other == run))
        value.x += ox, value.y += oy, value.z += oz;
    }
    if (head && (vertex != kInvalid) && ((value.x != 0.0f) || (value.y != 0.0f) || (value.z != 0.0f)))
      film_add(p, p.camera_sum + film_index(it, pixel), value * it.vm_normalization);
  }
  block_stat_add(p, kBlockStatMerged, merged_count, &s_stat);
}

// before the first camera bounce of an iteration: an empty histogram (afterwards k_merge_diffuse leaves one behind)
void launch_merge_reset(hipStream_t stream, const Pipeline& p) {
  (void)hipMemsetAsync(p.merge_buckets, 0, (kMergeBuckets + 1u) * sizeof(uint32_t), stream);
}

void launch_merge(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool generic_materials, uint32_t max_items) {
  const uint32_t vertex_blocks = max(1u, grid_for(min(max_items, p.capacity)));
  hipLaunchKernelGGL(k_merge_scan_totals, dim3(kMergeScanBlocks), dim3(kBlockSize), 0, stream, p);
  hipLaunchKernelGGL(k_merge_scan, dim3(kMergeScanBlocks), dim3(kBlockSize), 0, stream, p);
  hipLaunchKernelGGL(k_merge_scatter, dim3(vertex_blocks), dim3(kBlockSize), 0, stream, p);
  // a multiple of 8 workgroups: one eighth of the sorted list per XCD
  const uint32_t blocks = max(8u, (grid_for(uint32_t(min(uint64_t(min(max_items, p.capacity)) * 8ull, 0xffffff00ull))) + 7u) & ~7u);
  hipLaunchKernelGGL(k_merge_diffuse, dim3(blocks), dim3(kBlockSize), 0, stream, p, it);
  if (generic_materials) {
    hipLaunchKernelGGL(k_merge_filter_generic, dim3(blocks), dim3(kBlockSize), 0, stream, p, it);
    // the pair count is on the device: a persistent grid over the list, as large as the list can be
    hipLaunchKernelGGL(k_merge_eval_generic, dim3(max(1u, grid_for(p.pair_capacity))), dim3(kBlockSize), 0, stream, p, it);
  }
}

}  // namespace etxd
