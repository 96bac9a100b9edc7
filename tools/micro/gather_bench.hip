// gather_bench.hip - how fast a CU gathers records through its vector L1 when every lane reads a DIFFERENT record.
//
// The tree traversal, pair-connection and merge kernels read 96-128-byte records (tree node, light vertex, photon) at per-lane addresses:
// a wave-instruction then touches up to 64 cache lines. This measures lane-loads per clock and CU for dword / dwordx2 / dwordx4 gathers as a
// function of (a) how many loads a lane issues per record (consecutive 16-byte pieces of one 128-byte record: 1, 2, 4, 8), (b) the table
// size (L1- / L2- / MALL-resident), (c) whether the lanes of a wave read different records or the same one, and the LDS equivalent
// (ds_read_b128 at per-lane addresses). Independent loads: the index chain does not depend on the loaded data (throughput, not latency).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bench tools/micro/gather_bench.hip && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ inline uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// kPieces float4 loads of one 128-byte record per iteration and lane; records = table size in 128-byte records (power of two)
template <int kPieces, bool kUniform>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, uint32_t records, uint32_t iters, float* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t key = kUniform ? (tid >> 6) : tid;  // one record per wave, or one per lane
  float acc = 0.0f;
#pragma unroll 2
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t r = hash_u32(key * 0x9e3779b9u + i) & (records - 1u);
    const float4* rec = table + size_t(r) * 8u;
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const float4 v = rec[p];
      acc += v.x + v.w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

// every lane of a wave picks one of `distinct` records (the records differ from instruction to instruction): what a gather into a SMALL table
// costs - a material or emitter table of a few entries, where many lanes of a wave want the same record
__global__ __launch_bounds__(256) void k_gather_few(const float4* __restrict__ table, uint32_t records, uint32_t distinct, uint32_t iters, float* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t wave = tid >> 6;
  float acc = 0.0f;
#pragma unroll 2
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t pick = hash_u32(tid * 0x9e3779b9u + i) % distinct;            // which of the wave's records this lane wants
    const uint32_t r = hash_u32((wave * 64u + pick) * 0x85ebca6bu + i) & (records - 1u);
    const float4 v = table[size_t(r) * 8u];
    acc += v.x + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int kDwords>
__global__ __launch_bounds__(256) void k_gather_narrow(const float* __restrict__ table, uint32_t records, uint32_t iters, float* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
#pragma unroll 2
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t r = hash_u32(tid * 0x9e3779b9u + i) & (records - 1u);
    const float* rec = table + size_t(r) * 32u;
    if (kDwords == 1) {
      acc += rec[0];
    } else {
      const float2 v = *reinterpret_cast<const float2*>(rec);
      acc += v.x + v.y;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

// the same from LDS: 256 records of 128 bytes (32 KB) staged per workgroup, per-lane ds_read_b128
template <int kPieces>
__global__ __launch_bounds__(256) void k_gather_lds(const float4* __restrict__ table, uint32_t iters, float* sink) {
  __shared__ float4 s_table[256 * 8];
  for (uint32_t i = threadIdx.x; i < 256u * 8u; i += 256u)
    s_table[i] = table[i];
  __syncthreads();
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
#pragma unroll 2
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t r = hash_u32(tid * 0x9e3779b9u + i) & 255u;
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const float4 v = s_table[r * 8u + p];
      acc += v.x + v.w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

// dword / dwordx2 gathers from LDS (ds_read_b32 / b64 at per-lane addresses)
template <int kDwords>
__global__ __launch_bounds__(256) void k_gather_lds_narrow(const float4* __restrict__ table, uint32_t iters, float* sink) {
  __shared__ float s_table[256 * 32];
  for (uint32_t i = threadIdx.x; i < 256u * 32u; i += 256u)
    s_table[i] = reinterpret_cast<const float*>(table)[i];
  __syncthreads();
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
#pragma unroll 2
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t r = hash_u32(tid * 0x9e3779b9u + i) & 8191u;  // any dword of the 32 KB
    if (kDwords == 1) {
      acc += s_table[r];
    } else {
      const float2 v = *reinterpret_cast<const float2*>(s_table + (r & ~1u));
      acc += v.x + v.y;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

// a small table staged in LDS and read through a GENERIC pointer (flat_load: the address decides between the LDS and the global path) - what code
// written against `const T*` tables costs when a kernel redirects a table pointer to its LDS copy without retyping the code. kDwords 1 / 2 / 4.
template <int kDwords>
__global__ __launch_bounds__(256) void k_gather_flat_lds(const float4* __restrict__ table, uint32_t iters, float* sink) {
  __shared__ float s_table[256 * 32];
  for (uint32_t i = threadIdx.x; i < 256u * 32u; i += 256u)
    s_table[i] = reinterpret_cast<const float*>(table)[i];
  __syncthreads();
  const float* generic = s_table;
  asm volatile("" : "+v"(generic));  // the compiler no longer knows the address space
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
#pragma unroll 2
  for (uint32_t i = 0; i < iters; ++i) {
    const uint32_t r = hash_u32(tid * 0x9e3779b9u + i) & 8191u;
    if (kDwords == 1) {
      acc += generic[r];
    } else if (kDwords == 2) {
      const float2 v = *reinterpret_cast<const float2*>(generic + (r & ~1u));
      acc += v.x + v.y;
    } else {
      const float4 v = *reinterpret_cast<const float4*>(generic + (r & ~3u));
      acc += v.x + v.w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

// no load at all: the cost of the index arithmetic of the loops above
__global__ __launch_bounds__(256) void k_no_load(uint32_t iters, float* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
#pragma unroll 2
  for (uint32_t i = 0; i < iters; ++i)
    acc += hash_u32(tid * 0x9e3779b9u + i) & 8191u;
  if (acc == 123456u) sink[0] = float(acc);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double clock_ghz = prop.clockRate * 1.0e-6;
  const int cus = prop.multiProcessorCount;
  const size_t table_bytes = size_t(1) << 30;
  float4* table;
  float* sink;
  hipMalloc(&table, table_bytes);
  hipMalloc(&sink, 64);
  hipMemset(table, 0, table_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const uint32_t blocks = uint32_t(cus) * 8u, iters = 256;
  auto report = [&](const char* name, int pieces, int bytes_per_load, float ms, size_t table_kb) {
    const double lane_loads = double(blocks) * 256.0 * iters * pieces;
    const double per_clk_cu = lane_loads / (ms * 1.0e-3 * clock_ghz * 1.0e9 * cus);
    printf("%-34s table %8zu KB  %d x %2d B per record: %7.3f ms  %6.2f lane-loads/clk/CU  %7.1f G lane-loads/s  %6.2f TB/s\n", name, table_kb, pieces, bytes_per_load, ms, per_clk_cu,
      lane_loads / ms * 1.0e-6, lane_loads * bytes_per_load / ms * 1.0e-9);
  };
#define RUN(KERNEL, NAME, PIECES, BYTES, RECORDS, ...)                                         \
  for (int rep = 0; rep < 2; ++rep) {                                                           \
    hipEventRecord(e0);                                                                         \
    hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(256), 0, 0, __VA_ARGS__);                     \
    hipEventRecord(e1);                                                                         \
    hipEventSynchronize(e1);                                                                    \
    float ms = 0;                                                                               \
    hipEventElapsedTime(&ms, e0, e1);                                                           \
    if (rep == 1) report(NAME, PIECES, BYTES, ms, size_t(RECORDS) * 128u / 1024u);              \
  }
  printf("%s, %d CUs, %.2f GHz, %u workgroups of 256, %u records per lane\n", prop.name, cus, clock_ghz, blocks, iters);
  for (uint32_t records : {16u, 128u, 1024u, 8192u, 65536u, 1u << 20, 1u << 23}) {  // 2 KB, 16 KB, 128 KB, 1 MB, 8 MB, 128 MB, 1 GB
    RUN((k_gather<1, false>), "dwordx4, record per lane", 1, 16, records, table, records, iters, sink)
    RUN((k_gather<2, false>), "dwordx4, record per lane", 2, 16, records, table, records, iters, sink)
    RUN((k_gather<4, false>), "dwordx4, record per lane", 4, 16, records, table, records, iters, sink)
    RUN((k_gather<8, false>), "dwordx4, record per lane", 8, 16, records, table, records, iters, sink)
    RUN((k_gather<8, true>), "dwordx4, record per WAVE", 8, 16, records, table, records, iters, sink)
    RUN((k_gather_narrow<1>), "dword, record per lane", 1, 4, records, reinterpret_cast<const float*>(table), records, iters, sink)
    RUN((k_gather_narrow<2>), "dwordx2, record per lane", 1, 8, records, reinterpret_cast<const float*>(table), records, iters, sink)
  }
  for (uint32_t distinct : {1u, 2u, 4u, 8u, 16u, 32u}) {
    char name[64];
    snprintf(name, sizeof(name), "dwordx4, %u records per wave", distinct);
    RUN((k_gather_few), name, 1, 16, 1024, table, 1024u, distinct, iters, sink)
  }
  RUN((k_no_load), "no load (index arithmetic only)", 1, 0, 0, iters, sink)
  RUN((k_gather_lds_narrow<1>), "LDS ds_read_b32, dword per lane", 1, 4, 256, table, iters, sink)
  RUN((k_gather_lds_narrow<2>), "LDS ds_read_b64, pair per lane", 1, 8, 256, table, iters, sink)
  RUN((k_gather_flat_lds<1>), "LDS via flat_load_dword", 1, 4, 256, table, iters, sink)
  RUN((k_gather_flat_lds<2>), "LDS via flat_load_dwordx2", 1, 8, 256, table, iters, sink)
  RUN((k_gather_flat_lds<4>), "LDS via flat_load_dwordx4", 1, 16, 256, table, iters, sink)
  RUN((k_gather_lds<1>), "LDS ds_read_b128, record per lane", 1, 16, 256, table, iters, sink)
  RUN((k_gather_lds<8>), "LDS ds_read_b128, record per lane", 8, 16, 256, table, iters, sink)
  return 0;
}
