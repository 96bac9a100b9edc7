// kernels_grid.hip - photon hash grid construction on the device.
// Restates VCMSpatialGrid::construct (sources/etx/rt/integrators/vcm_shared.cxx:49-152):
//   bbox of the mergeable light vertices -> hash table of next_pow2(N) cells, cell size 2r ->
//   count per cell (atomics) -> exclusive prefix sum (the reference scans serially, :109-114) ->
//   scatter into a cell-sorted SoA with an atomic cursor per cell (afterwards cell_ends[c] = end of cell c).
// Roofline: HBM (streaming N x 80 B in, N x 64 B out, 4 B x table size for count/scan).
#include "kernels.h"
#include "dev_vcm.h"

namespace etxd {

constexpr uint32_t kScanItemsPerThread = 8;
constexpr uint32_t kScanBlockItems = kBlockSize * kScanItemsPerThread;  // 2048

ETX_DEV float ordered_to_float(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

ETX_DEV uint32_t next_power_of_two(uint32_t v) {  // math.hxx:1012-1021
  v--;
  v |= v >> 1u, v |= v >> 2u, v |= v >> 4u, v |= v >> 8u, v |= v >> 16u;
  return v + 1u;
}

ETX_DEV uint32_t float_to_ordered(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// bounding box of the mergeable light vertices (vcm_shared.cxx:66-80): wave shuffles, then 6 atomics per wave
__global__ __launch_bounds__(kBlockSize) void k_grid_bbox(Pipeline p) {
  const uint32_t n = min(p.counters[kCntLightVertices], p.lv.capacity);
  f3 lo = mk3(kMaxFloat), hi = mk3(-kMaxFloat);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (__float_as_uint(p.lv.nrm_tri(i).w) == kInvalid)
      continue;
    float4 pd = p.lv.pos_dvcm(i);
    lo = fmin3(lo, f3{pd.x, pd.y, pd.z});
    hi = fmax3(hi, f3{pd.x, pd.y, pd.z});
  }
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    lo.x = fminf(lo.x, __shfl_xor(lo.x, d)), lo.y = fminf(lo.y, __shfl_xor(lo.y, d)), lo.z = fminf(lo.z, __shfl_xor(lo.z, d));
    hi.x = fmaxf(hi.x, __shfl_xor(hi.x, d)), hi.y = fmaxf(hi.y, __shfl_xor(hi.y, d)), hi.z = fmaxf(hi.z, __shfl_xor(hi.z, d));
  }
  // block level: the 4 waves meet in LDS, one lane per block issues the 6 atomics (they all hit the same 6 words)
  __shared__ float s_box[kBlockSize / 64][6];
  if ((threadIdx.x & 63u) == 0u) {
    float* b = s_box[threadIdx.x >> 6];
    b[0] = lo.x, b[1] = lo.y, b[2] = lo.z, b[3] = hi.x, b[4] = hi.y, b[5] = hi.z;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t w = 1; w < kBlockSize / 64; ++w) {
      lo = fmin3(lo, f3{s_box[w][0], s_box[w][1], s_box[w][2]});
      hi = fmax3(hi, f3{s_box[w][3], s_box[w][4], s_box[w][5]});
    }
  }
  if ((threadIdx.x == 0) && (lo.x <= hi.x)) {
    atomicMin(p.counters + kCntBboxMin + 0, float_to_ordered(lo.x));
    atomicMin(p.counters + kCntBboxMin + 1, float_to_ordered(lo.y));
    atomicMin(p.counters + kCntBboxMin + 2, float_to_ordered(lo.z));
    atomicMax(p.counters + kCntBboxMax + 0, float_to_ordered(hi.x));
    atomicMax(p.counters + kCntBboxMax + 1, float_to_ordered(hi.y));
    atomicMax(p.counters + kCntBboxMax + 2, float_to_ordered(hi.z));
  }
}

__global__ void k_grid_setup(Pipeline p, VcmParams it) {
  if ((blockIdx.x != 0) || (threadIdx.x != 0))
    return;
  GridParams g;
  uint32_t sample_count = min(p.counters[kCntLightVertices], p.lv.capacity);
  float radius = it.current_radius;
  g.radius_squared = radius * radius;
  g.inv_radius_squared = (g.radius_squared > 0.0f) ? 1.0f / g.radius_squared : 0.0f;
  g.cell_size = 2.0f * radius;
  g.bbox_min = {ordered_to_float(p.counters[kCntBboxMin + 0]), ordered_to_float(p.counters[kCntBboxMin + 1]), ordered_to_float(p.counters[kCntBboxMin + 2])};
  g.bbox_max = {ordered_to_float(p.counters[kCntBboxMax + 0]), ordered_to_float(p.counters[kCntBboxMax + 1]), ordered_to_float(p.counters[kCntBboxMax + 2])};
  uint32_t table = sample_count ? next_power_of_two(sample_count) : 1u;
  if (table > p.grid.hash_capacity)
    table = p.grid.hash_capacity;  // hash_capacity is a power of two >= light vertex capacity
  g.hash_mask = table - 1u;
  g.photon_count = 0u;  // filled by k_grid_finish
  g.valid = (sample_count > 0u) && (p.counters[kCntBboxMax + 0] != 0u) ? 1u : 0u;
  *p.grid_params = g;
}

__global__ __launch_bounds__(kBlockSize) void k_grid_clear(Pipeline p) {
  const uint32_t n = p.grid_params->hash_mask + 1u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    p.grid.cell_ends[i] = 0u;
}

__global__ __launch_bounds__(kBlockSize) void k_grid_count(Pipeline p) {
  const GridParams g = *p.grid_params;
  if (g.valid == 0u)
    return;
  const uint32_t n = min(p.counters[kCntLightVertices], p.lv.capacity);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 nt = p.lv.nrm_tri(i);
    if (__float_as_uint(nt.w) == kInvalid)
      continue;  // medium vertices are never merged (vcm_shared.cxx:100-102)
    float4 pd = p.lv.pos_dvcm(i);
    atomicAdd(p.grid.cell_ends + grid_position_to_index(g, f3{pd.x, pd.y, pd.z}), 1u);
  }
}

// exclusive scan, pass 1: every block scans 2048 cells in LDS and publishes its total
__global__ __launch_bounds__(kBlockSize) void k_scan_blocks(Pipeline p) {
  __shared__ uint32_t s_wave_sums[kBlockSize / 64];
  const uint32_t n = p.grid_params->hash_mask + 1u;
  const uint32_t block_count = (n + kScanBlockItems - 1u) / kScanBlockItems;
  for (uint32_t block = blockIdx.x; block < block_count; block += gridDim.x) {
    const uint32_t first = block * kScanBlockItems + threadIdx.x * kScanItemsPerThread;
    uint32_t v[kScanItemsPerThread];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanItemsPerThread; ++k) {
      v[k] = (first + k < n) ? p.grid.cell_ends[first + k] : 0u;
      sum += v[k];
    }
    // wave inclusive scan of the per-thread sums
    uint32_t incl = sum;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(incl, d);
      if ((threadIdx.x & 63u) >= d)
        incl += t;
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 63u)
      s_wave_sums[wave] = incl;
    __syncthreads();
    uint32_t wave_offset = 0;
    for (uint32_t w = 0; w < wave; ++w)
      wave_offset += s_wave_sums[w];
    uint32_t running = wave_offset + incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < kScanItemsPerThread; ++k) {
      if (first + k < n)
        p.grid.cell_ends[first + k] = running;
      running += v[k];
    }
    if (threadIdx.x == kBlockSize - 1u)
      p.grid.block_sums[block] = running;
    __syncthreads();
  }
}

// pass 2: one block scans the block totals (<= hash_capacity / 2048 entries)
__global__ __launch_bounds__(1024) void k_scan_sums(Pipeline p) {
  __shared__ uint32_t s_part[1024];
  const uint32_t n = p.grid_params->hash_mask + 1u;
  const uint32_t block_count = (n + kScanBlockItems - 1u) / kScanBlockItems;
  const uint32_t per_thread = (block_count + 1023u) / 1024u;
  const uint32_t first = threadIdx.x * per_thread;
  uint32_t sum = 0;
  for (uint32_t k = 0; k < per_thread; ++k)
    sum += (first + k < block_count) ? p.grid.block_sums[first + k] : 0u;
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024u; d <<= 1) {
    uint32_t t = (threadIdx.x >= d) ? s_part[threadIdx.x - d] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t running = s_part[threadIdx.x] - sum;
  for (uint32_t k = 0; k < per_thread; ++k) {
    if (first + k < block_count) {
      uint32_t v = p.grid.block_sums[first + k];
      p.grid.block_sums[first + k] = running;
      running += v;
    }
  }
  if (threadIdx.x == 1023u)
    p.grid_params->photon_count = s_part[1023];
}

// pass 3: add the block offsets
__global__ __launch_bounds__(kBlockSize) void k_scan_add(Pipeline p) {
  const uint32_t n = p.grid_params->hash_mask + 1u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    p.grid.cell_ends[i] += p.grid.block_sums[i / kScanBlockItems];
}

__global__ __launch_bounds__(kBlockSize) void k_grid_scatter(Pipeline p) {
  const GridParams g = *p.grid_params;
  if (g.valid == 0u)
    return;
  const uint32_t n = min(p.counters[kCntLightVertices], p.lv.capacity);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 nt = p.lv.nrm_tri(i);
    if (__float_as_uint(nt.w) == kInvalid)
      continue;
    float4 pd = p.lv.pos_dvcm(i);
    float4 wd = p.lv.wi_dvc(i);
    float4 td = p.lv.thr_dvm(i);
    float4 bl = p.lv.bc_len_med(i);
    uint32_t cell = grid_position_to_index(g, f3{pd.x, pd.y, pd.z});
    uint32_t dst = atomicAdd(p.grid.cell_ends + cell, 1u);
    p.grid.pos_len[dst] = make_float4(pd.x, pd.y, pd.z, __uint_as_float(__float_as_uint(bl.z) & 0xffffu));
    p.grid.nrm_dvcm(dst) = make_float4(nt.x, nt.y, nt.z, pd.w);
    p.grid.win_dvm(dst) = make_float4(wd.x, wd.y, wd.z, td.w);
    // vcm_shared.cxx:140 + vcm_shared.hxx:873-877: (throughput / sampling_pdf).to_rgb(), in spectral mode scaled by
    // SpectralDistribution::kRGBLuminanceScale; the merge multiplies it channel-wise with the camera side's RGB
    f3 thr = {td.x, td.y, td.z};
    if (p.scene.spectral)
      thr = thr * spectral_film_weight(p.scene, p.lv.wavelength(i)) * f3{0.817660332f, 1.05418909f, 1.09945524f};
    p.grid.thr(dst) = make_float4(thr.x, thr.y, thr.z, 0.0f);
  }
}

void launch_grid_build(hipStream_t stream, const Pipeline& p, const VcmParams& it) {
  const uint32_t blocks = min(kPersistentBlocks, (p.capacity + kBlockSize - 1) / kBlockSize);
  hipLaunchKernelGGL(k_grid_bbox, dim3(512), dim3(kBlockSize), 0, stream, p);
  hipLaunchKernelGGL(k_grid_setup, dim3(1), dim3(64), 0, stream, p, it);
  hipLaunchKernelGGL(k_grid_clear, dim3(kPersistentBlocks), dim3(kBlockSize), 0, stream, p);
  hipLaunchKernelGGL(k_grid_count, dim3(kPersistentBlocks), dim3(kBlockSize), 0, stream, p);
  hipLaunchKernelGGL(k_scan_blocks, dim3(kPersistentBlocks), dim3(kBlockSize), 0, stream, p);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, p);
  hipLaunchKernelGGL(k_scan_add, dim3(kPersistentBlocks), dim3(kBlockSize), 0, stream, p);
  hipLaunchKernelGGL(k_grid_scatter, dim3(kPersistentBlocks), dim3(kBlockSize), 0, stream, p);
  (void)blocks;
}

}  // namespace etxd
