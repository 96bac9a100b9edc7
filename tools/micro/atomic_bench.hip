#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// every wave performs `iters` returning atomics (lane 0) on counter[(wave_id % spread) * stride]
__global__ void k_atomic(uint32_t* counters, uint32_t iters, uint32_t spread, uint32_t stride, uint32_t* sink) {
  uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63u;
  uint32_t acc = 0;
  uint32_t* c = counters + (wave % spread) * stride;
  for (uint32_t i = 0; i < iters; ++i) {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(c, 1u);
    v = __shfl(v, 0);
    acc += v;
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}
__global__ void k_atomic_noret(uint32_t* counters, uint32_t iters, uint32_t spread, uint32_t stride) {
  uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63u;
  uint32_t* c = counters + (wave % spread) * stride;
  for (uint32_t i = 0; i < iters; ++i)
    if (lane == 0) atomicAdd(c, 1u);
}
// float atomics scattered like film splats: each lane adds to pixel (base + lane) * 4 floats
__global__ void k_film(float* film, uint32_t iters, uint32_t pixels) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = 0; i < iters; ++i) {
    uint32_t px = (tid + i * stride) % pixels;
    atomicAdd(film + px * 4 + 0, 1.0f);
    atomicAdd(film + px * 4 + 1, 1.0f);
    atomicAdd(film + px * 4 + 2, 1.0f);
  }
}
int main() {
  uint32_t* counters; uint32_t* sink; float* film;
  hipMalloc(&counters, 1 << 20); hipMalloc(&sink, 64); hipMalloc(&film, 1920 * 1080 * 16);
  hipMemset(counters, 0, 1 << 20); hipMemset(film, 0, 1920 * 1080 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t blocks = 2048, iters = 64;
  const double total = double(blocks) * 4 * iters;
  struct Cfg { uint32_t spread, stride; const char* name; } cfgs[] = {
    {1, 1, "one address"}, {3, 1, "3 addresses, same line"}, {3, 32, "3 addresses, 128 B apart"}, {8, 32, "8 addresses 128 B apart"}, {64, 32, "64 addresses"}, {8192, 32, "private address per wave"}};
  for (auto& c : cfgs) {
    for (int ret = 1; ret >= 0; --ret) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (ret) hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, counters, iters, c.spread, c.stride, sink);
        else hipLaunchKernelGGL(k_atomic_noret, dim3(blocks), dim3(256), 0, 0, counters, iters, c.spread, c.stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf("%-28s %s: %.3f ms  %.1f M atomics/s  (%.1f ns each, chip-wide)\n", c.name, ret ? "returning" : "no return", ms, total / ms / 1e3, ms * 1e6 / total);
    }
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_film, dim3(2048), dim3(256), 0, 0, film, 16u, 1920u * 1080u);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("film float atomics: %.3f ms for %.1f M atomics -> %.1f G/s\n", ms, 2048.0 * 256 * 16 * 3 / 1e6, 2048.0 * 256 * 16 * 3 / ms / 1e6);
  return 0;
}
