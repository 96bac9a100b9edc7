// graph_bench.hip - is a hipGraph replay of a planned VCM pass cheaper than the stream launches it would replace?
//
// A planned pass (csrc/host_api.cpp run_bounce_loop, scheduled branch) is a fixed CHAIN of launches on one stream: per round a traversal kernel,
// a shade kernel, a shadow kernel (+ connect / merge kernels in the camera pass), ~440 launches per iteration, every kernel taking a ~250-byte
// pipeline header and a ~60-byte per-iteration parameter block BY VALUE (Pipeline, VcmParams) - the iteration index, the merge radius and the launch
// bounds change every iteration, so a replayed graph needs its kernel-node parameters rewritten before every replay.
// This models exactly that: kChain dependent launches of a kernel with a 320-byte by-value argument that does a few microseconds of work on a
// buffer (grid of `blocks` workgroups), four ways:
//   stream        hipLaunchKernelGGL x kChain on a stream (what the library does today)
//   graph         one hipGraphLaunch of the captured chain, parameters untouched (the floor: no per-iteration update)
//   graph+params  hipGraphExecKernelNodeSetParams on EVERY node, then hipGraphLaunch (what a per-iteration replay needs)
//   graph+update  re-capture the chain into a new graph, hipGraphExecUpdate the instantiated one, hipGraphLaunch (the other way to change arguments)
// Reported per chain: host time to enqueue (the lane thread's cost), wall time until the stream is idle, device time between first and last
// kernel (HIP events), for 1 and 4 concurrent streams (the library keeps four iterations in flight, one host thread each).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/graph_bench tools/micro/graph_bench.hip -lpthread && tools/micro/bin/graph_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      fprintf(stderr, "%s failed: %s (line %d)\n", #call, hipGetErrorString(e_), __LINE__);          \
      exit(1);                                                                                       \
    }                                                                                                \
  } while (0)

struct Params {  // stands in for Pipeline + VcmParams: 320 bytes by value
  float* buffer;
  uint32_t count;
  uint32_t iteration;
  float radius;
  uint32_t pad[75];
};
static_assert(sizeof(Params) == 320, "by-value argument block");

__global__ __launch_bounds__(256) void k_round(Params p) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.count; i += gridDim.x * blockDim.x)
    p.buffer[i] = p.buffer[i] * 0.999f + p.radius + float(p.iteration & 1u);
}

constexpr int kChain = 440;   // launches of one VCM iteration on configs[1]
constexpr int kRepeats = 40;  // chains per measurement

struct Lane {
  hipStream_t stream;
  float* buffer;
  uint32_t count, blocks;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<hipGraphNode_t> nodes;
  hipEvent_t begin, end;
  double host_ms = 0.0, wall_ms = 0.0, device_ms = 0.0;
};

static void enqueue_chain(Lane& l, uint32_t iteration) {
  Params p = {};
  p.buffer = l.buffer, p.count = l.count, p.iteration = iteration, p.radius = 1.0f / float(iteration + 1u);
  for (int k = 0; k < kChain; ++k)
    hipLaunchKernelGGL(k_round, dim3(l.blocks), dim3(256), 0, l.stream, p);
}

static void capture(Lane& l, uint32_t iteration, hipGraph_t* out) {
  CHECK(hipStreamBeginCapture(l.stream, hipStreamCaptureModeThreadLocal));
  enqueue_chain(l, iteration);
  CHECK(hipStreamEndCapture(l.stream, out));
}

enum Mode { kStream, kGraph, kGraphParams, kGraphUpdate };

static void run(Lane& l, Mode mode) {
  using clock = std::chrono::steady_clock;
  double host = 0.0, device = 0.0;
  const auto wall0 = clock::now();
  for (int r = 0; r < kRepeats; ++r) {
    const uint32_t iteration = uint32_t(r);
    const auto t0 = clock::now();
    CHECK(hipEventRecord(l.begin, l.stream));
    if (mode == kStream) {
      enqueue_chain(l, iteration);
    } else if (mode == kGraph) {
      CHECK(hipGraphLaunch(l.exec, l.stream));
    } else if (mode == kGraphParams) {
      Params p = {};
      p.buffer = l.buffer, p.count = l.count, p.iteration = iteration, p.radius = 1.0f / float(iteration + 1u);
      void* args[] = {&p};
      hipKernelNodeParams np = {};
      np.func = reinterpret_cast<void*>(k_round);
      np.gridDim = dim3(l.blocks), np.blockDim = dim3(256), np.sharedMemBytes = 0, np.kernelParams = args, np.extra = nullptr;
      for (hipGraphNode_t node : l.nodes)
        CHECK(hipGraphExecKernelNodeSetParams(l.exec, node, &np));
      CHECK(hipGraphLaunch(l.exec, l.stream));
    } else {
      hipGraph_t fresh = nullptr;
      capture(l, iteration, &fresh);
      hipGraphNode_t error_node = nullptr;
      hipGraphExecUpdateResult result;
      CHECK(hipGraphExecUpdate(l.exec, fresh, &error_node, &result));
      CHECK(hipGraphDestroy(fresh));
      CHECK(hipGraphLaunch(l.exec, l.stream));
    }
    CHECK(hipEventRecord(l.end, l.stream));
    host += std::chrono::duration<double, std::milli>(clock::now() - t0).count();
    CHECK(hipEventSynchronize(l.end));  // one chain in flight per lane, like one iteration per lane
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, l.begin, l.end));
    device += ms;
  }
  l.wall_ms = std::chrono::duration<double, std::milli>(clock::now() - wall0).count() / kRepeats;
  l.host_ms = host / kRepeats, l.device_ms = device / kRepeats;
}

int main() {
  CHECK(hipSetDevice(0));
  const char* names[] = {"stream", "graph", "graph+params", "graph+update"};
  for (uint32_t blocks : {64u, 512u}) {          // a thin late round / a full early round
    const uint32_t count = blocks * 256u * 4u;
    for (int lanes : {1, 4}) {
      std::vector<Lane> ls(lanes);
      for (Lane& l : ls) {
        CHECK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        CHECK(hipMalloc(reinterpret_cast<void**>(&l.buffer), count * sizeof(float)));
        CHECK(hipMemset(l.buffer, 0, count * sizeof(float)));
        l.count = count, l.blocks = blocks;
        CHECK(hipEventCreate(&l.begin));
        CHECK(hipEventCreate(&l.end));
        capture(l, 0u, &l.graph);
        size_t n = 0;
        CHECK(hipGraphGetNodes(l.graph, nullptr, &n));
        l.nodes.resize(n);
        CHECK(hipGraphGetNodes(l.graph, l.nodes.data(), &n));
        CHECK(hipGraphInstantiate(&l.exec, l.graph, nullptr, nullptr, 0));
      }
      for (int mode = 0; mode < 4; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {  // second pass is the measurement
          std::vector<std::thread> threads;
          for (Lane& l : ls)
            threads.emplace_back([&l, mode] { CHECK(hipSetDevice(0)); run(l, Mode(mode)); });
          for (auto& t : threads)
            t.join();
        }
        double host = 0.0, wall = 0.0, device = 0.0;
        for (Lane& l : ls)
          host += l.host_ms / lanes, wall += l.wall_ms / lanes, device += l.device_ms / lanes;
        printf("%3u workgroups x %d launches, %d lane(s), %-13s host enqueue %7.3f ms   chain wall %7.3f ms   device first..last %7.3f ms   (%.2f us / launch wall)\n", blocks, kChain, lanes,
               names[mode], host, wall, device, wall * 1000.0 / kChain);
      }
      for (Lane& l : ls) {
        CHECK(hipGraphExecDestroy(l.exec));
        CHECK(hipGraphDestroy(l.graph));
        CHECK(hipFree(l.buffer));
        CHECK(hipEventDestroy(l.begin));
        CHECK(hipEventDestroy(l.end));
        CHECK(hipStreamDestroy(l.stream));
      }
    }
  }
  return 0;
}
