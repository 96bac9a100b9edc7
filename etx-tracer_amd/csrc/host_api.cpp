// host_api.cpp - the C ABI of libetx_hip.so (include/etx_hip.h): context, device pools, the per-iteration launch
// sequence that stands in for CPUVCMImpl::start_next_iteration / gather_* / complete_* (vcm_cpu.cxx:95-241).
#include "../../include/etx_hip.h"

#include "host_scene.h"
#include "host_reduce.h"
#include "host_transfer.h"
#include "kernels_bvh_build.h"
#include "kernels.h"
#include "dev_bvh.h"
#include "tuning_knobs.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace etxd;

#if defined(ETX_HIP_DEBUG_COUNTERS)
namespace etxd { void launch_debug_lists(hipStream_t stream, const Pipeline& p); }
static void launch_debug_lists_fwd(hipStream_t s, const etxd::Pipeline& p) { etxd::launch_debug_lists(s, p); }
#endif

namespace {

thread_local std::string g_create_error;

enum TimerId : uint32_t {
  kTimerTraceClosest = 0,
  kTimerTraceShadow,
  kTimerShadeLight,
  kTimerShadeCamera,
  kTimerConnect,
  kTimerMerge,
  kTimerGridBuild,
  kTimerGenerate,
  kTimerCount,
};

struct TimedSpan {
  hipEvent_t begin, end;
  uint32_t id;
};

}  // namespace

constexpr uint32_t kFilmLayers = 4;     // camera, light, normal, albedo sums (pipeline.h)
constexpr uint32_t kBlueNoiseSets = 9;  // sample-count classes 1, 2, 4, ... 256 of the host's blue-noise sampler

struct etx_hip_context;
int etx_hip_internal_reduce_reset(etx_hip_context* c);
int etx_hip_internal_reduce_allocate(etx_hip_context* c);
void etx_hip_internal_reduce_release(etx_hip_context* c);

struct etx_hip_context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  etxh::DeviceScene scene;
  Pipeline pipe = {};
  std::vector<void*> allocations;
  // The per-iteration pools whose fill depends on the scene and the sample (light vertices + photon grid, camera vertices, pairs, shadow and
  // endpoint queues) start at what typical paths need and GROW: an iteration that overflows one is not committed (k_vcm_commit), the pool is
  // doubled and the iteration rendered again (execute_iteration). `pool_sizes` = what this lane holds, `pool_wanted` (public context) =
  // what every lane adopts before its next iteration; sizes only grow between two etx_hip_upload_scene / etx_hip_update_scene calls.
  struct PoolSizes {
    uint32_t light_vertices = 0, pairs = 0, shadow = 0, camera_vertices = 0, endpoints = 0;
    bool operator==(const PoolSizes& o) const {
      return (light_vertices == o.light_vertices) && (pairs == o.pairs) && (shadow == o.shadow) && (camera_vertices == o.camera_vertices) && (endpoints == o.endpoints);
    }
  };
  PoolSizes pool_sizes, pool_wanted;
  std::vector<void*> pool_allocations;
  size_t pool_bytes = 0;             // of this lane's growable pools (part of allocated_bytes)
  bool retry_attempt = false;        // execute_iteration: the iteration being rendered was discarded once (pool overflow)
  bool grid_wanted = false;          // the photon grid belongs to the pools once a VCM run has begun (allocate_photon_grid)
  uint32_t pool_initial_per_path = 0;  // etx_hip_set_pool_policy: light vertices per path the pools start with (0 = default), public context
  size_t pool_limit_bytes = 0;       // ... and the size a lane's pools may grow to (0 = no limit but the device's memory), public context
  bool scene_ready = false;
  bool armed = false;
  int integrator = ETX_HIP_INTEGRATOR_VCM;
  etx_abi_vcm_options vcm_options = {};
  etx_abi_pt_options pt_options = {};
  etx_abi_bdpt_options bdpt_options = {};
  float noise_threshold = 0.0f;       // path tracing: Scene::noise_threshold when adaptive sampling is active for this run, else 0
  float4* adaptive_sum = nullptr;     // public context: Pipeline::adaptive_sum / pixel_state storage (allocated at the first adaptive run)
  uint32_t* pixel_state = nullptr;
  size_t adaptive_pixels = 0;
  float4* pt_iteration_image = nullptr;  // camera and light contributions of the iteration this lane renders (2 x pixels), committed to the film at its end
  uint32_t first_iteration = 0, iteration_stride = 1;
  uint32_t pixel_first = 0, pixel_stride = 1;  // etx_hip_begin_ex: this context's share of the pixels (path tracer, bidirectional integrator)
  uint32_t pool_paths = 0;                     // paths per sub pass pool_wanted was first sized for
  uint32_t next_iteration = 0;       // iteration index to render next
  uint32_t local_iterations = 0;     // iterations rendered by this context since begin
  uint64_t global_iterations = 0;    // after a film reduce: iterations of all ranks at the time of that reduce
  EtxReduceState reduce;             // public context: the multi-GPU film reduce (host_reduce.h, host_comm.cpp)
  hipEvent_t commit_done = nullptr;  // recorded behind this lane's newest commit kernel, under the public context's reduce.mutex
  bool commit_recorded = false;
  uint32_t tail_divisor = 32;        // active paths <= capacity / tail_divisor: finish the pass in the tail kernel (0 = never); configs[1]: 64 -> 94.2, 32 -> 96.3, 16 -> 95.2 Msamples/s
  uint32_t check_interval = 3;       // rounds the host may enqueue beyond the newest round the device has reported (run_bounce_loop)
  uint32_t timer_mask = (1u << kTimerTraceClosest) | (1u << kTimerTraceShadow);
  uint32_t cross_mode = 0;           // which path state the traversal kernel may advance across medium boundaries (kernels.h launch_trace_closest): set per iteration by the integrator
  size_t allocated_bytes = 0;        // of this lane's pipeline (etx_hip_device_bytes)
  uint32_t base_lanes = 1, active_lanes = 1;  // public context: lanes every integrator uses / lanes the armed integrator uses
  uint32_t max_lanes = 1;                     // ... / lanes the bidirectional integrator uses: those beyond base_lanes (thread, stream, events, pools) are created by its first etx_hip_begin
  uint32_t debug_flags = 0;          // etx_hip_set_debug_flags: ablation switches of the kernels (Pipeline::debug_flags), 0 in production
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
  std::vector<TimedSpan> spans;
  hipEvent_t iteration_begin = nullptr, iteration_end = nullptr;
  uint32_t* host_counters = nullptr;  // pinned
  unsigned long long* round_mirror = nullptr;  // pinned: (round tag + 1) << 32 | active paths, written by k_trace_closest (run_bounce_loop)
  uint32_t next_round_tag = 0;               // never reset: stale entries of earlier passes cannot match
  // The launch plan of a pass (run_bounce_loop): live paths entering each round of this lane's previous iteration, read from the round mirror
  // after the iteration. With a plan the host enqueues the whole pass - every round with the plan's launch shape, then the tail kernel -
  // without waiting for the device in between; the kernels take their true counts from the device counters as always.
  struct PassPlan {
    std::vector<uint32_t> entering;
    uint32_t first_tag = 0, rounds_enqueued = 0;  // of the pass being / last rendered: where its mirror entries are
  };
  PassPlan plans[2];                         // light pass, camera pass (path tracing: [1] only)
  bool scheduled_passes = true;              // ETX_HIP_SCHEDULED_PASSES=0 (debug builds): always poll
  uint32_t iterations_on_plan = 0;           // iterations since this lane last polled its passes (execute_iteration drops the plans every kPlanLifetime)
  uint8_t* bluenoise[kBlueNoiseSets] = {};  // device tables by sample-count class (etx_hip_upload_bluenoise)
  const uint8_t* active_bluenoise = nullptr;
  float4* cie_table = nullptr;      // spectrum::spectral_xyz (etx_hip_upload_cie_table), spectral scenes only
  float4* rgb_response_table = nullptr;  // rgb_response rows (etx_hip_upload_rgb_response), spectral scenes with RGB images
  uint32_t rgb_response_count = 0;
  float rgb_response_first = 0.0f;
  uint32_t cie_count = 0;
  float cie_first = 0.0f, cie_y_scale = 0.0f;
  etx_hip_stats_t stats = {};
  float4* resolve_buffer = nullptr;
  // asynchronous film read-back (etx_hip_read_film_begin / _end): own stream, device resolve buffer, pinned staging
  hipStream_t read_stream = nullptr;
  hipEvent_t read_event = nullptr;
  float4* read_resolve = nullptr;
  float4* read_staging = nullptr;
  size_t read_pixels = 0;
  bool read_pending = false;
  void* comm = nullptr;  // ncclComm_t (host_comm.cpp)
  etxh::HostTransfer transfer;  // public context: every host <-> device copy of caller-owned memory goes through its pinned slots (host_transfer.h)
  int rank = 0, world = 1;

  // Asynchronous execution. A context is a set of LANES: the public context itself plus helper contexts, each with its
  // own stream, per-iteration pools, counters and worker thread; all lanes add into the film of the public context
  // (float atomics). etx_hip_render_iteration hands the next iteration to a free lane and returns, so consecutive
  // iterations overlap on the device: while one is in its thin tail (a few thousand long paths, launch-latency bound)
  // the wide first bounces of the next one fill the CUs.
  etx_hip_context* owner = nullptr;       // helper lanes: the public context
  std::vector<etx_hip_context*> helpers;  // public context: its helper lanes (owned)
  std::thread worker;
  std::mutex lane_mutex;                  // jobs / quit of this lane
  std::condition_variable lane_cv;
  std::deque<uint32_t> jobs;              // iteration indices handed to this lane
  bool lane_busy = false;                 // guarded by the owner's shared_mutex
  bool quit = false;
  // public context only
  std::mutex shared_mutex;
  std::condition_variable idle_cv;
  uint32_t jobs_in_flight = 0;
  int sticky_error = 0;
  std::string sticky_error_text;
  etx_hip_stats_t totals = {};            // since etx_hip_begin
  std::chrono::steady_clock::time_point busy_since;

  bool fail(int, const std::string& msg) {
    error = msg;
    return false;
  }
};

namespace {

#define HIP_OK(ctx, call)                                                                 \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (ctx)->error = std::string(#call) + " failed: " + hipGetErrorString(e_);           \
      return ETX_HIP_ERROR_HIP;                                                           \
    }                                                                                     \
  } while (0)

#define TRANSFER_OK(ctx, call)                \
  do {                                        \
    if (int rc_ = (call)) {                   \
      (void)(ctx);                            \
      return rc_;                             \
    }                                         \
  } while (0)

template <class T>
int device_alloc(etx_hip_context* ctx, T*& ptr, size_t count) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    ctx->error = "hipMalloc(" + std::to_string(bytes) + ") failed: " + hipGetErrorString(e);
    return ETX_HIP_ERROR_HIP;
  }
  ctx->allocations.push_back(p);
  ctx->allocated_bytes += bytes;
  ptr = reinterpret_cast<T*>(p);
  return 0;
}

template <class T>
int pool_alloc(etx_hip_context* ctx, T*& ptr, size_t count) {
  const size_t before = ctx->allocated_bytes;
  if (int rc = device_alloc(ctx, ptr, count))
    return rc;
  ctx->pool_allocations.push_back(ctx->allocations.back());
  ctx->allocations.pop_back();
  ctx->pool_bytes += ctx->allocated_bytes - before;
  return 0;
}

void release_pools(etx_hip_context* ctx) {
  for (void* p : ctx->pool_allocations)
    (void)hipFree(p);
  ctx->pool_allocations.clear();
  ctx->allocated_bytes -= std::min(ctx->allocated_bytes, ctx->pool_bytes);
  ctx->pool_bytes = 0;
  ctx->pool_sizes = {};
  // nothing may keep pointing at freed memory: every pool pointer and capacity of the pipeline is cleared with the pools
  Pipeline& p = ctx->pipe;
  p.grid.cell_ends = nullptr, p.grid.pos_len = nullptr, p.grid.rec = nullptr, p.grid.block_sums = nullptr;
  p.lv.rec = nullptr, p.lv.capacity = 0u;
  p.cv = {}, p.merge_order = nullptr, p.cv_capacity = 0u;
  p.pairs = nullptr, p.pair_capacity = 0u;
  p.path_chunks = nullptr, p.path_chunk_capacity = 0u;
  p.shadow = {};
  p.endpoints = {};
  // the lane's own medium table (scenes whose walks append rows) went with the pools: back to the shared one
  p.scene.mediums = ctx->scene.host_copy.mediums;
  p.scene.dyn_medium_first = p.scene.dyn_medium_capacity = 0u;
}

void release_pipeline(etx_hip_context* ctx) {
  release_pools(ctx);
  ctx->grid_wanted = false;
  for (void* p : ctx->allocations)
    (void)hipFree(p);
  ctx->allocations.clear();
  ctx->allocated_bytes = 0;
  ctx->pipe = {};
  ctx->resolve_buffer = nullptr;
  ctx->pt_iteration_image = nullptr;
  ctx->adaptive_sum = nullptr, ctx->pixel_state = nullptr, ctx->adaptive_pixels = 0;
}

uint32_t next_pow2(uint32_t v) {
  v--;
  v |= v >> 1, v |= v >> 2, v |= v >> 4, v |= v >> 8, v |= v >> 16;
  return v + 1;
}

constexpr uint32_t kPoolRecordLimit = 1u << 30;

// What the pools start with. The reference grows a std::vector of light vertices under a mutex (vcm_cpu.cxx:131-171); measured at 1080p the
// bench scenes store 2.9-3.8 vertices per light path (fog box 3.8, classic 3.5, gems 3.4, density-grid box 2.9), the subsurface scene of
// configs[3] 9.7 (one per scattering event of a walk). Round 3 sized for 16 / 64 per path and for sixteen pairs per camera vertex of a
// bounce - 9.2 GB per lane on a 44-triangle box, 185 GB for configs[3] on six lanes. Now: 5 (16 with subsurface materials) per path,
// as many pairs per bounce, and whatever a scene needs beyond that is found by the overflow / retry path.
etx_hip_context::PoolSizes initial_pool_sizes(const etx_hip_context* pub, uint32_t n) {
  etx_hip_context::PoolSizes sizes;
  const bool sss = pub->scene.has_subsurface;
  uint32_t per_path = sss ? 16u : 5u;
  if (pub->pool_initial_per_path != 0u)
    per_path = pub->pool_initial_per_path;
  sizes.light_vertices = uint32_t(std::min<uint64_t>(uint64_t(n) * per_path, kPoolRecordLimit));
  // camera vertex records per path and bounce: Christensen-Burley vertices have up to 24 exit points; a subsurface walk under the
  // bidirectional integrator stores its entry and its exit vertex in one kernel invocation
  const uint32_t exit_points = pub->scene.has_subsurface_cb ? 8u : (sss ? 2u : 1u);
  sizes.camera_vertices = uint32_t(std::min<uint64_t>(uint64_t(n) * exit_points, kPoolRecordLimit));
  sizes.pairs = uint32_t(std::min<uint64_t>(uint64_t(n) * std::max(per_path, 4u), kPoolRecordLimit));
  sizes.shadow = uint32_t(std::min<uint64_t>(uint64_t(sizes.pairs) + 2ull * sizes.camera_vertices, 0xfffffff0ull));
  // endpoint requests come from the general / subsurface shading groups only (k_connect_endpoints)
  sizes.endpoints = (pub->scene.group_general || pub->scene.group_subsurface) ? sizes.camera_vertices : 1024u;
  return sizes;
}

size_t pool_bytes_for(const etx_hip_context::PoolSizes& z, bool with_grid) {
  size_t bytes = size_t(z.light_vertices) * LightVertexPool::kLvStride * sizeof(float4);
  if (with_grid)
    bytes += size_t(next_pow2(z.light_vertices)) * 4u + size_t(z.light_vertices) * (1u + PhotonGrid::kPhotonStride) * sizeof(float4);
  bytes += size_t(z.camera_vertices) * (7u * sizeof(float4) + 3u * 4u);
  bytes += size_t(z.pairs) * sizeof(uint2) + size_t(z.shadow) * 3u * sizeof(float4) + size_t(z.endpoints) * (5u * sizeof(float4) + 4u);
  bytes += (size_t(z.light_vertices) / kPathChunkEntries + size_t(z.light_vertices) / (kPathTableEntries - kBdptRowHeader + 1u) + 1u) * kPathChunkWords * sizeof(uint32_t);  // path_chunks
  return bytes;
}

// (Re)allocates this lane's growable pools at `sizes`; the lane's stream must be idle.
int allocate_pools(etx_hip_context* ctx, const etx_hip_context::PoolSizes& sizes) {
  release_pools(ctx);
  Pipeline& p = ctx->pipe;
  int rc = 0;
  p.lv.capacity = sizes.light_vertices;
  if ((rc = pool_alloc(ctx, p.lv.rec, size_t(p.lv.capacity) * LightVertexPool::kLvStride)))
    return rc;
  p.grid.hash_capacity = next_pow2(p.lv.capacity);
  if (ctx->grid_wanted) {
    if ((rc = pool_alloc(ctx, p.grid.cell_ends, p.grid.hash_capacity)) || (rc = pool_alloc(ctx, p.grid.pos_len, p.lv.capacity)) ||
        (rc = pool_alloc(ctx, p.grid.rec, size_t(p.lv.capacity) * PhotonGrid::kPhotonStride)) || (rc = pool_alloc(ctx, p.grid.block_sums, p.grid.hash_capacity / 2048u + 1024u)))
      return rc;
  }
  p.cv_capacity = sizes.camera_vertices;
  const uint32_t cvn = p.cv_capacity;
  if ((rc = pool_alloc(ctx, p.cv.wavelength, cvn)) || (rc = pool_alloc(ctx, p.cv.hit, cvn)) || (rc = pool_alloc(ctx, p.cv.wi_medium, cvn)) || (rc = pool_alloc(ctx, p.cv.thr_depth, cvn)) ||
      (rc = pool_alloc(ctx, p.cv.mis_pixel, cvn)) || (rc = pool_alloc(ctx, p.cv.seed, cvn)) || (rc = pool_alloc(ctx, p.cv.pos_info, cvn)) || (rc = pool_alloc(ctx, p.cv.nrm_dvm, cvn)) ||
      (rc = pool_alloc(ctx, p.cv.fthr_dvcm, cvn)) || (rc = pool_alloc(ctx, p.merge_order, cvn)))
    return rc;
  p.pair_capacity = sizes.pairs;
  if ((rc = pool_alloc(ctx, p.pairs, p.pair_capacity)))
    return rc;
  // overflow chunks of the bidirectional light paths' index lists (pipeline.h kBdptRowHeader): a path beyond its row takes one chunk per 31 vertices and leaves at most one
  // partly filled, and only a path of more vertices than its row holds takes any - lv / 31 + lv / (row entries + 1) chunks can never run out before the vertex pool does
  p.path_chunk_capacity = p.lv.capacity / kPathChunkEntries + p.lv.capacity / (p.path_table_entries - kBdptRowHeader + 1u) + 1u;
  if ((rc = pool_alloc(ctx, p.path_chunks, size_t(p.path_chunk_capacity) * kPathChunkWords)))
    return rc;
  p.shadow.capacity = sizes.shadow;
  if ((rc = pool_alloc(ctx, p.shadow.p0_medium, p.shadow.capacity)) || (rc = pool_alloc(ctx, p.shadow.p1_target, p.shadow.capacity)) || (rc = pool_alloc(ctx, p.shadow.value, p.shadow.capacity)))
    return rc;
  p.endpoints.capacity = sizes.endpoints;
  const uint32_t epn = sizes.endpoints;
  if ((rc = pool_alloc(ctx, p.endpoints.hit, epn)) || (rc = pool_alloc(ctx, p.endpoints.wi_medium, epn)) || (rc = pool_alloc(ctx, p.endpoints.thr_depth, epn)) ||
      (rc = pool_alloc(ctx, p.endpoints.mis_id, epn)) || (rc = pool_alloc(ctx, p.endpoints.rnd_seed, epn)) || (rc = pool_alloc(ctx, p.endpoints.wavelength, epn)))
    return rc;
  // Textured subsurface materials under the bidirectional integrator (DScene::sss_dynamic_media): the lane's own copy of the medium table with room for
  // one row per walk behind it - as many as the light vertex pool holds records (a walk stores at least its entry vertex there or in the camera pool,
  // and an overflow of either grows all pools)
  p.scene.lane_counters = p.counters;
  if (ctx->scene.host_copy.sss_dynamic_media != 0u) {
    const uint32_t rows = ctx->scene.medium_table_rows;
    DMedium* table = nullptr;
    if ((rc = pool_alloc(ctx, table, size_t(rows) + size_t(sizes.light_vertices))))
      return rc;
    HIP_OK(ctx, hipMemcpyAsync(table, ctx->scene.host_copy.mediums, size_t(rows) * sizeof(DMedium), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    p.scene.mediums = table;
    p.scene.dyn_medium_first = rows, p.scene.dyn_medium_capacity = sizes.light_vertices;
  }
  ctx->pool_sizes = sizes;
  return 0;
}

int allocate_pipeline(etx_hip_context* ctx) {
  release_pipeline(ctx);
  Pipeline& p = ctx->pipe;
  const uint32_t n = ctx->scene.film_w * ctx->scene.film_h;
  p.scene = ctx->scene.host_copy;
  // a tree whose stack bound exceeds the LDS part: this lane's spill area (dev_bvh.h LaneStack), one column per thread of the
  // largest grid any traversing kernel is launched with
  p.scene.stack_spill = nullptr, p.scene.stack_spill_lanes = 0u;
  if ((p.scene.bvh_flat == 0u) && (p.scene.bvh_stack_need > kShortStackDepth)) {  // rows for the kernels with the short LDS stack, which cover the others'
    const uint32_t spill_lanes = 2u * kPersistentBlocks * kBlockSize;
    // rows: what THIS tree can need beyond the short LDS stack (its exact bound from the host; at least the 48 rows every tree got until round 5)
    const uint32_t spill_rows = std::max(p.scene.bvh_stack_need, 64u) - kShortStackDepth;
    if (int rc = device_alloc(ctx, p.scene.stack_spill, size_t(spill_lanes) * spill_rows))
      return rc;
    p.scene.stack_spill_lanes = spill_lanes;
  }
  p.debug_flags = ctx->debug_flags;
  p.capacity = n;
  int rc = 0;
  for (int s = 0; s < 2; ++s) {
    if ((rc = device_alloc(ctx, p.paths[s].ray_o_tmin, n)) || (rc = device_alloc(ctx, p.paths[s].ray_d_tmax, n)) || (rc = device_alloc(ctx, p.paths[s].thr_eta, n)) ||
        (rc = device_alloc(ctx, p.paths[s].mis, n)) || (rc = device_alloc(ctx, p.paths[s].meta, n)) || (rc = device_alloc(ctx, p.paths[s].path_id, n)) || (rc = device_alloc(ctx, p.paths[s].wavelength, n)) ||
        (rc = device_alloc(ctx, p.paths[s].prev_pos, n)) || (rc = device_alloc(ctx, p.paths[s].prev_nrm, n)))
      return rc;
  }
  if ((rc = device_alloc(ctx, p.hits, n)))
    return rc;
  p.walk[0] = p.walk[1] = p.walk_exit = {}, p.walk_info[0] = p.walk_info[1] = nullptr, p.walk_exit_hits = nullptr;
  if (ctx->scene.has_subsurface) {  // walk and exit queues of the bidirectional integrator (k_bdpt_walk_*)
    PathSet* sets[3] = {&p.walk[0], &p.walk[1], &p.walk_exit};
    for (PathSet* e : sets) {
      if ((rc = device_alloc(ctx, e->ray_o_tmin, n)) || (rc = device_alloc(ctx, e->ray_d_tmax, n)) || (rc = device_alloc(ctx, e->thr_eta, n)) || (rc = device_alloc(ctx, e->mis, n)) ||
          (rc = device_alloc(ctx, e->meta, n)) || (rc = device_alloc(ctx, e->path_id, n)) || (rc = device_alloc(ctx, e->wavelength, n)) || (rc = device_alloc(ctx, e->prev_pos, n)) ||
          (rc = device_alloc(ctx, e->prev_nrm, n)))
        return rc;
    }
    if ((rc = device_alloc(ctx, p.walk_info[0], n)) || (rc = device_alloc(ctx, p.walk_info[1], n)) || (rc = device_alloc(ctx, p.walk_exit_hits, n)))
      return rc;
  }
  if ((rc = device_alloc(ctx, p.path_wavelength, n)))
    return rc;
  // a light path that walks through a subsurface object under the bidirectional integrator stores a vertex per scattering event: a longer
  // table keeps k_expand_pairs off the per-lane list walk (configs[3]: 181 us per launch with eight entries)
  p.path_table_entries = std::max(8u, etxh::tuning_knob("ETX_HIP_PATH_TABLE", ctx->scene.has_subsurface ? kPathTableEntriesWalk : kPathTableEntries) & ~3u);  // (>= 8: the weights of BDPTFast read a path's first two entries from its row)
  if ((rc = device_alloc(ctx, p.light_path_table, size_t(n) * (p.path_table_entries / 4u))))
    return rc;
  p.grid = {};
  if ((rc = device_alloc(ctx, p.grid_params, 1)))
    return rc;
  if ((rc = device_alloc(ctx, p.group_list[0], n)) || (rc = device_alloc(ctx, p.group_list[1], n)))
    return rc;
  if ((rc = device_alloc(ctx, p.merge_buckets, kMergeBuckets + 1u + 256u)))
    return rc;
  // the growable pools (light vertices, camera vertices, pairs, shadow and endpoint queues; the photon grid from the first VCM run on)
  {
    etx_hip_context* pub = ctx->owner ? ctx->owner : ctx;
    if (pub->pool_wanted.light_vertices == 0u) {
      pub->pool_wanted = initial_pool_sizes(pub, n);
      pub->pool_paths = n;
    }
    if ((rc = allocate_pools(ctx, pub->pool_wanted)))
      return rc;
  }
  if ((rc = device_alloc(ctx, ctx->pt_iteration_image, size_t(n) * 2u)))
    return rc;
  if (ctx->owner == nullptr) {
    if ((rc = device_alloc(ctx, p.camera_sum, size_t(n) * kFilmLayers)) || (rc = device_alloc(ctx, ctx->resolve_buffer, n)))
      return rc;
  } else {
    p.camera_sum = ctx->owner->pipe.camera_sum;  // every lane adds into the public context's film
  }
  if ((rc = device_alloc(ctx, p.block_stats, kBlockStatRows * kBlockStatCount)))
    return rc;
  if ((rc = device_alloc(ctx, p.counters, kCounterCount)))
    return rc;
  HIP_OK(ctx, hipMemset(p.counters, 0, kCounterCount * sizeof(uint32_t)));
  p.scene.lane_counters = p.counters;  // (allocate_pools ran before the counters existed)
  HIP_OK(ctx, hipMemset(p.grid_params, 0, sizeof(GridParams)));
  p.light_sum = p.camera_sum + n;
  p.normal_sum = p.camera_sum + 2u * size_t(n);
  p.albedo_sum = p.camera_sum + 3u * size_t(n);
  if (ctx->owner == nullptr)
    HIP_OK(ctx, hipMemset(p.camera_sum, 0, size_t(n) * kFilmLayers * sizeof(float4)));
  HIP_OK(ctx, hipMemset(ctx->pt_iteration_image, 0, size_t(n) * 2u * sizeof(float4)));
  return 0;
}

// VCMSpatialGrid storage of one lane (80 B per pooled light vertex), on first use (etx_hip_begin with the VCM integrator): from then on it is
// part of the lane's growable pools
int allocate_photon_grid(etx_hip_context* ctx) {
  Pipeline& p = ctx->pipe;
  ctx->grid_wanted = true;
  if (p.grid.rec != nullptr)
    return 0;
  HIP_OK(ctx, hipSetDevice(ctx->device));
  int rc = 0;
  if ((rc = pool_alloc(ctx, p.grid.cell_ends, p.grid.hash_capacity)) || (rc = pool_alloc(ctx, p.grid.pos_len, p.lv.capacity)) ||
      (rc = pool_alloc(ctx, p.grid.rec, size_t(p.lv.capacity) * PhotonGrid::kPhotonStride)) || (rc = pool_alloc(ctx, p.grid.block_sums, p.grid.hash_capacity / 2048u + 1024u)))
    return rc;
  return 0;
}

constexpr int kBaseLanes = 4, kBidirectionalLanes = 6;

bool lane_ready(const etx_hip_context* lane) {
  return lane->allocations.empty() == false;
}

uint32_t lanes_for_integrator(const etx_hip_context* context, int integrator) {
  return (integrator == ETX_HIP_INTEGRATOR_BDPT) ? context->max_lanes : std::min(context->max_lanes, context->base_lanes);
}

// pools of the lanes every integrator uses, after the public context's own (etx_hip_upload_scene / etx_hip_update_scene)
int allocate_base_lanes(etx_hip_context* context) {
  for (size_t i = 0; i + 1u < context->base_lanes && i < context->helpers.size(); ++i) {
    etx_hip_context* helper = context->helpers[i];
    helper->scene.borrow(context->scene);
    if (int rc = allocate_pipeline(helper)) {
      context->error = helper->error;
      return rc;
    }
  }
  return 0;
}

hipEvent_t take_event(etx_hip_context* ctx) {
  if (ctx->events_used == ctx->event_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess)
      return nullptr;
    ctx->event_pool.push_back(e);
  }
  return ctx->event_pool[ctx->events_used++];
}

// A lane's commit kernel (iteration image -> film sums) and the snapshot of a film reduce exclude each other on the DEVICE (host_reduce.h):
// the commit is enqueued behind the newest snapshot, and its own end is recorded for the next snapshot to wait for - both under the public
// context's reduce.mutex, so every snapshot holds whole iterations. No host thread waits; a lane's stream stalls only if its commit falls
// into the ~30 us a snapshot kernel runs.
struct CommitSection {
  etx_hip_context* lane;
  EtxReduceState& reduce;
  explicit CommitSection(etx_hip_context* l)
    : lane(l)
    , reduce((l->owner ? l->owner : l)->reduce) {
    reduce.mutex.lock();
    if (reduce.snapshot_recorded)
      (void)hipStreamWaitEvent(lane->stream, reduce.snapshot_done, 0);
  }
  ~CommitSection() {
    if (lane->commit_done != nullptr) {
      (void)hipEventRecord(lane->commit_done, lane->stream);
      lane->commit_recorded = true;
    }
    reduce.mutex.unlock();
  }
};

struct ScopedTimer {
  etx_hip_context* ctx;
  bool active;
  TimedSpan span;
  ScopedTimer(etx_hip_context* c, uint32_t id)
    : ctx(c)
    , active((c->timer_mask >> id) & 1u) {
    if (active) {
      span.id = id;
      span.begin = take_event(ctx);
      span.end = take_event(ctx);
      active = span.begin && span.end;
      if (active)
        (void)hipEventRecord(span.begin, ctx->stream);
    }
  }
  ~ScopedTimer() {
    if (active) {
      (void)hipEventRecord(span.end, ctx->stream);
      ctx->spans.push_back(span);
    }
  }
};

int read_counters(etx_hip_context* ctx) {
  HIP_OK(ctx, hipMemcpyAsync(ctx->host_counters, ctx->pipe.counters, kCounterCount * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// pixels first, first + stride, ... below n
uint32_t shard_paths(uint32_t n, uint32_t first, uint32_t stride) {
  return (first < n) ? (n - first + stride - 1u) / stride : 0u;
}

// vcm_cpu.cxx:95-124 start_next_iteration: radius schedule and VC/VM weights
VcmParams make_iteration_params(const etx_hip_context* ctx, uint32_t iteration) {
  const auto& o = ctx->vcm_options;
  const auto& sc = ctx->scene.host_copy;
  VcmParams it = {};
  it.options = (o.options & ETX_VCM_FULL_OPTIONS) | (o.reference_seeding ? kOptionReferenceSeeding : 0u);
  it.kernel = o.kernel;
  it.iteration = iteration;
  it.film_w = ctx->scene.film_w, it.film_h = ctx->scene.film_h;
  it.path_count = it.film_w * it.film_h;
  it.pixel_first = 0u, it.pixel_stride = 1u;  // a photon map holds the light paths of every pixel: VCM is never pixel-sharded (etx_hip_begin_ex)
  float used_radius = o.initial_radius;
  if (used_radius == 0.0f) {
    uint32_t max_dim = std::max(it.film_w, it.film_h);
    used_radius = 5.0f * sc.bounds_radius / float(max_dim);
  }
  float radius_scale = 1.0f / (1.0f + float(iteration) / float(o.radius_decay));
  it.current_radius = used_radius * radius_scale;
  float eta_vcm = kPi * it.current_radius * it.current_radius * float(it.path_count);
  it.vc_weight = 1.0f / eta_vcm;
  it.vm_weight = (o.options & ETX_VCM_ENABLE_MERGING) ? eta_vcm : 0.0f;
  it.vm_normalization = 1.0f / eta_vcm;
  it.bluenoise = reinterpret_cast<const uint2*>(ctx->active_bluenoise);
  return it;
}

ShadeGroups shade_groups(const etx_hip_context* ctx) {
  ShadeGroups g;
  g.general = ctx->scene.group_general;
  g.subsurface = ctx->scene.group_subsurface;
  return g;
}

// One pass of the wavefront loop: trace + shade rounds until no path is alive. The active count lives on the device.
// The trace kernel of every round mirrors (round tag, active paths entering the round) into pinned host memory; the
// host enqueues rounds ahead of the device (at most `run_ahead` rounds beyond the newest mirrored one) and reads the
// mirror without ever draining the stream, so the device does not idle while the host learns that the pass has ended.
// Rounds enqueued after the last path died are empty launches (at most `run_ahead` of them).
template <class ShadeFn, class TailFn>
int run_bounce_loop(etx_hip_context* ctx, ShadeFn&& shade, TailFn&& tail, uint64_t& rounds, uint32_t pass_stat, bool allow_tail = true) {
  uint32_t set = 0;
  uint32_t known_count = ctx->pipe.capacity;  // upper bound of the active paths (the count never grows within a pass)
  const uint32_t tail_threshold = (allow_tail && ctx->tail_divisor) ? std::max(64u, ctx->pipe.capacity / ctx->tail_divisor) : 0u;
  etx_hip_context::PassPlan& plan = ctx->plans[(pass_stat == kStatRaysLight) ? 0 : 1];
  const std::vector<uint32_t> planned = plan.entering;
  plan.first_tag = ctx->next_round_tag, plan.rounds_enqueued = 0u;
  // A pass that ends in the tail kernel needs no answer from the device at all: the rounds of the plan (launch shapes of this lane's last
  // iteration: which round follows which is fixed, how many paths a round finds is on the device), then the tail kernel, which runs whatever
  // is alive to its end - fewer or more paths than the plan expected cost time, not correctness. The host comes back at the end of the
  // iteration and reads the round mirror for the next plan.
  if (ctx->scheduled_passes && allow_tail && (tail_threshold != 0u) && (planned.empty() == false) && (planned.size() + 2u < kRoundMirrorSlots)) {
    uint32_t bound = ctx->pipe.capacity;
    for (size_t r = 0; r < planned.size(); ++r) {
      if (planned[r] <= tail_threshold)
        break;
      bound = uint32_t(std::min<uint64_t>(ctx->pipe.capacity, uint64_t(planned[r]) + planned[r] / 8u + 4096u));
      const uint32_t tag = ctx->next_round_tag++;
      {
        ScopedTimer t(ctx, kTimerTraceClosest);
        launch_trace_closest(ctx->stream, ctx->pipe, set, set == 0 ? kCntActiveA : kCntActiveB, bound, ctx->scene.host_copy.bvh_flat != 0u, ctx->round_mirror, tag, pass_stat, ctx->cross_mode);
      }
      shade(set, bound);
      set ^= 1u;
      rounds++;
      plan.rounds_enqueued++;
    }
    tail(set, bound);
    rounds++;
    // a scheduled pass never reads the device's answers: one query per pass keeps a device fault from surfacing only at the iteration's end event
    const hipError_t q = hipStreamQuery(ctx->stream);
    if ((q != hipSuccess) && (q != hipErrorNotReady)) {
      ctx->error = std::string("wavefront loop (scheduled pass): ") + hipGetErrorString(q);
      return ETX_HIP_ERROR_HIP;
    }
    return 0;
  }
  // Alive paths are bounded by depth plus roulette, but boundary crossings do not add depth: the loop runs until the
  // device reports no active path (a path that never ends would be a defect, reported as an error - never a silent cut).
  const uint64_t max_rounds = uint64_t(ctx->scene.host_copy.max_path_length) * 64ull + 4096ull;
  const uint32_t run_ahead = std::max(1u, std::min(ctx->check_interval, kRoundMirrorSlots / 2u));
  const uint32_t first_tag = ctx->next_round_tag;
  uint32_t newest_seen = first_tag;  // tag of the oldest round whose mirror entry has not been read yet
  volatile unsigned long long* mirror = ctx->round_mirror;
  auto poll = [&](uint32_t enqueued_until) {  // consumes the mirror entries that have arrived; returns the newest count
    while (newest_seen != enqueued_until) {
      const unsigned long long entry = mirror[newest_seen & (kRoundMirrorSlots - 1u)];
      if (uint32_t(entry >> 32u) != newest_seen + 1u)
        break;
      known_count = uint32_t(entry & 0xffffffffull);
      newest_seen += 1u;
    }
  };
  for (uint64_t round = 0; round < max_rounds; ++round) {
    const uint32_t tag = ctx->next_round_tag++;
    {
      ScopedTimer t(ctx, kTimerTraceClosest);
      launch_trace_closest(ctx->stream, ctx->pipe, set, set == 0 ? kCntActiveA : kCntActiveB, known_count, ctx->scene.host_copy.bvh_flat != 0u, ctx->round_mirror, tag, pass_stat, ctx->cross_mode);
    }
    shade(set, known_count);
    set ^= 1u;
    rounds++;
    plan.rounds_enqueued++;
    // wait until the device is at most `run_ahead` rounds behind what has been enqueued
    // (a few hundred polls cover the rounds of a busy pass; after that the thread sleeps between polls instead of keeping a host core
    // at 100 % per lane, and asks the stream for errors about once per millisecond)
    uint32_t spins = 0;
    for (poll(tag + 1u); (tag + 1u) - newest_seen > run_ahead; poll(tag + 1u)) {
      spins += 1u;
      const bool dozing = spins > 256u;
      if (dozing ? (((spins - 256u) & 31u) == 0u) : ((spins & 127u) == 0u)) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if ((q != hipSuccess) && (q != hipErrorNotReady)) {
          ctx->error = std::string("wavefront loop: ") + hipGetErrorString(q);
          return ETX_HIP_ERROR_HIP;
        }
        if (q == hipSuccess) {  // stream idle: every mirror entry must have arrived
          poll(tag + 1u);
          if ((tag + 1u) - newest_seen > run_ahead) {
            ctx->error = "wavefront loop: the device finished without reporting its rounds";
            return ETX_HIP_ERROR_STATE;
          }
        }
      }
      if (dozing)
        std::this_thread::sleep_for(std::chrono::microseconds(20));
      else
        std::this_thread::yield();
    }
    if (newest_seen == first_tag)
      continue;  // nothing known yet
    if (known_count == 0u)
      return 0;  // the rounds still in flight are empty
    if (known_count <= tail_threshold) {
      tail(set, known_count);
      rounds++;
      return 0;
    }
  }
  ctx->error = "wavefront loop: " + std::to_string(known_count) + " paths still alive after " + std::to_string(max_rounds) + " rounds";
  return ETX_HIP_ERROR_STATE;
}

int render_vcm_iteration(etx_hip_context* ctx, uint32_t iteration) {
  const VcmParams it = make_iteration_params(ctx, iteration);
  ctx->cross_mode = 1u;  // VCMPathState: medium in meta.z, path distance in mis.w
  // The kernels add into the lane's ITERATION images; k_vcm_commit folds them into the film sums at the end of the iteration
  // (Film::commit_light_iteration, film.cxx:332-343, and the per-iteration camera value of vcm_cpu.cxx:227-241). Adding every
  // connection / splat straight into sums that have grown over thousands of iterations would absorb the contributions that
  // are smaller than half an ulp of the sum - a negative bias that grows with the sample count (measured at 4096 spp).
  Pipeline p = ctx->pipe;
  p.camera_sum = ctx->pt_iteration_image;
  p.light_sum = ctx->pt_iteration_image + p.capacity;
  hipStream_t s = ctx->stream;
  uint64_t rounds = 0;

  launch_iteration_reset(s, p);
  {
    ScopedTimer t(ctx, kTimerGenerate);
    launch_light_generate(s, p, it);
  }
  int rc = run_bounce_loop(
    ctx,
    [&](uint32_t set, uint32_t max_items) {
      {
        ScopedTimer t(ctx, kTimerShadeLight);
        launch_light_shade(s, p, it, set, max_items, shade_groups(ctx));
        if (opt_connect_to_camera(it) && shade_groups(ctx).binned())
          launch_connect_endpoints(s, p, it, false, max_items);
      }
      if (opt_connect_to_camera(it)) {
        ScopedTimer t(ctx, kTimerTraceShadow);
        launch_trace_shadow(s, p, max_items, ctx->scene.host_copy.bvh_flat != 0u);
      }
    },
    [&](uint32_t set, uint32_t max_items) {
      {
        ScopedTimer t(ctx, kTimerShadeLight);
        launch_light_tail(s, p, it, set, max_items, shade_groups(ctx));
        if (opt_connect_to_camera(it) && shade_groups(ctx).binned())
          launch_connect_endpoints(s, p, it, false, p.endpoints.capacity);
      }
      if (opt_connect_to_camera(it)) {
        ScopedTimer t(ctx, kTimerTraceShadow);
        launch_trace_shadow(s, p, p.shadow.capacity, ctx->scene.host_copy.bvh_flat != 0u);
      }
    },
    rounds, kStatRaysLight);
  if (rc)
    return rc;

#if defined(ETX_HIP_DEBUG_COUNTERS)
  launch_debug_lists_fwd(s, p);
#endif
  if (opt_merge_vertices(it)) {
    ScopedTimer t(ctx, kTimerGridBuild);
    launch_grid_build(s, p, it);
  }

  {
    ScopedTimer t(ctx, kTimerGenerate);
    launch_camera_generate(s, p, it);
  }
  if (opt_merge_vertices(it))
    launch_merge_reset(s, p);
  rc = run_bounce_loop(
    ctx,
    [&](uint32_t set, uint32_t max_items) {
      {
        ScopedTimer t(ctx, kTimerShadeCamera);
        launch_camera_shade(s, p, it, set, max_items, shade_groups(ctx));
        if (opt_connect_to_light(it) && shade_groups(ctx).binned())
          launch_connect_endpoints(s, p, it, true, max_items);
      }
      if (opt_connect_vertices(it)) {
        ScopedTimer t(ctx, kTimerConnect);
        launch_connect(s, p, it, ctx->scene.generic_materials, max_items);
      }
      if (opt_connect_vertices(it) || opt_connect_to_light(it)) {
        ScopedTimer t(ctx, kTimerTraceShadow);
        launch_trace_shadow(s, p, uint32_t(std::min<uint64_t>(uint64_t(max_items) * 6ull, 0xffffffffull)), ctx->scene.host_copy.bvh_flat != 0u);
      }
      if (opt_merge_vertices(it)) {
        ScopedTimer t(ctx, kTimerMerge);
        launch_merge(s, p, it, ctx->scene.generic_materials, max_items);
      }
    },
    [&](uint32_t set, uint32_t max_items) {
      {
        ScopedTimer t(ctx, kTimerShadeCamera);
        launch_camera_tail(s, p, it, set, max_items, shade_groups(ctx));
        if (opt_connect_to_light(it) && shade_groups(ctx).binned())
          launch_connect_endpoints(s, p, it, true, p.endpoints.capacity);
      }
      // the tail leaves up to `capacity` camera vertices: drain them with one launch of each consumer
      if (opt_connect_vertices(it)) {
        ScopedTimer t(ctx, kTimerConnect);
        launch_connect(s, p, it, ctx->scene.generic_materials, p.capacity);
      }
      if (opt_connect_vertices(it) || opt_connect_to_light(it)) {
        ScopedTimer t(ctx, kTimerTraceShadow);
        launch_trace_shadow(s, p, p.shadow.capacity, ctx->scene.host_copy.bvh_flat != 0u);
      }
      if (opt_merge_vertices(it)) {
        ScopedTimer t(ctx, kTimerMerge);
        launch_merge(s, p, it, ctx->scene.generic_materials, p.capacity);
      }
    },
    rounds, kStatRaysCamera);
  if (rc)
    return rc;
  {
    CommitSection commit(ctx);
    launch_vcm_commit(s, p.camera_sum, p.light_sum, ctx->pipe.camera_sum, ctx->pipe.light_sum, p.capacity, p.counters);
  }
  launch_stats_finalize(s, p);
  ctx->stats.wavefront_bounces = rounds;
  return 0;
}

// One path-tracing iteration (CPUPathTracingImpl::execute_range over all pixels, path_tracing.cxx:50-83).
int render_pt_iteration(etx_hip_context* ctx, uint32_t iteration) {
  const auto& o = ctx->pt_options;
  VcmParams it = {};
  it.options = (o.direct ? ETX_PT_DIRECT : 0u) | (o.nee ? ETX_PT_NEE : 0u) | (o.mis ? ETX_PT_MIS : 0u);
  it.iteration = iteration;
  it.film_w = ctx->scene.film_w, it.film_h = ctx->scene.film_h;
  it.pixel_first = ctx->pixel_first, it.pixel_stride = ctx->pixel_stride;
  it.path_count = shard_paths(it.film_w * it.film_h, it.pixel_first, it.pixel_stride);
  it.bluenoise = reinterpret_cast<const uint2*>(ctx->active_bluenoise);
  ctx->cross_mode = 0u;
  hipStream_t s = ctx->stream;
  // the shade and shadow kernels add into the ITERATION image (radiance clamp applies to the iteration's pixel value)
  Pipeline p = ctx->pipe;
  p.camera_sum = ctx->pt_iteration_image;
  uint64_t rounds = 0;
  launch_iteration_reset(s, p);
  {
    ScopedTimer t(ctx, kTimerGenerate);
    launch_pt_generate(s, p, it);
  }
  const bool flat = ctx->scene.host_copy.bvh_flat != 0u;
  int rc = run_bounce_loop(
    ctx,
    [&](uint32_t set, uint32_t max_items) {
      {
        ScopedTimer t(ctx, kTimerShadeCamera);
        launch_pt_shade(s, p, it, set, max_items, shade_groups(ctx));
      }
      if (o.direct || o.nee) {
        ScopedTimer t(ctx, kTimerTraceShadow);
        launch_trace_shadow(s, p, uint32_t(std::min<uint64_t>(uint64_t(max_items) * 2ull, p.shadow.capacity)), flat);
      }
    },
    [&](uint32_t, uint32_t) {}, rounds, kStatRaysCamera, false);
  if (rc)
    return rc;
  {
    CommitSection commit(ctx);
    launch_pt_commit(s, ctx->pt_iteration_image, ctx->pipe.camera_sum, ctx->pipe.adaptive_sum, ctx->pipe.capacity, ctx->scene.host_copy.radiance_clamp);  // the whole frame: a pixel shard's pixels are spread over it
  }
  // Film::estimate_noise_levels(status.current_iteration, ...), path_tracing.cxx:99: after even iterations from kMinSamples = 32 on
  if ((ctx->pipe.pixel_state != nullptr) && (iteration >= 32u) && ((iteration & 1u) == 0u))
    launch_noise_estimate(s, ctx->pipe, it.film_w, it.film_h, ctx->noise_threshold);
  launch_stats_finalize(s, ctx->pipe);
  ctx->stats.wavefront_bounces = rounds;
  return 0;
}

// One bidirectional iteration (CPUBidirectionalImpl::execute_range over all pixels, bidirectional.cxx:352-403): the emitter
// sub paths of all pixels, then the camera sub paths; every connection's visibility goes through the shadow queue.
int render_bdpt_iteration(etx_hip_context* ctx, uint32_t iteration) {
  const auto& o = ctx->bdpt_options;
  VcmParams it = {};
  it.options = (o.connect_to_camera ? ETX_VCM_CONNECT_TO_CAMERA : 0u) | (o.direct_hit ? ETX_VCM_DIRECT_HIT : 0u) | (o.connect_to_light ? ETX_VCM_CONNECT_TO_LIGHT : 0u) |
               (o.connect_vertices ? ETX_VCM_CONNECT_VERTICES : 0u) | (o.mis ? ETX_VCM_ENABLE_MIS : 0u) | (o.reference_seeding ? kOptionReferenceSeeding : 0u);
  if (ctx->retry_attempt)
    it.options |= kOptionRetryKeepsAovs;  // the discarded first attempt has added this iteration's normal / albedo values
  it.kernel = o.mode;  // CPUBidirectionalImpl::Mode
  ctx->cross_mode = 2u;  // BdptState: medium in meta.z, no path distance
  it.iteration = iteration;
  it.film_w = ctx->scene.film_w, it.film_h = ctx->scene.film_h;
  it.pixel_first = ctx->pixel_first, it.pixel_stride = ctx->pixel_stride;
  it.path_count = shard_paths(it.film_w * it.film_h, it.pixel_first, it.pixel_stride);
  it.bluenoise = reinterpret_cast<const uint2*>(ctx->active_bluenoise);
  Pipeline p = ctx->pipe;  // iteration images, as in render_vcm_iteration
  p.camera_sum = ctx->pt_iteration_image;
  p.light_sum = ctx->pt_iteration_image + p.capacity;
  hipStream_t s = ctx->stream;
  const bool flat = ctx->scene.host_copy.bvh_flat != 0u;
  // the BSDF instantiations of the bidirectional kernels (kernels_bdpt.hip ETX_BDPT_LAUNCH)
  // (debug flag 0x10000: a mixed scene is not split by BSDF class - the general instantiations take every item, as until round 5; for A/B runs and tests/test_gpu_bdpt.py)
  const uint32_t simple = ctx->scene.simple_materials ? kBdptKernelsSimple : ((ctx->scene.bdpt_binning && ((ctx->debug_flags & 0x10000u) == 0u)) ? kBdptKernelsBinned : kBdptKernelsGeneral);
  uint64_t rounds = 0;
  int rc = 0;

  launch_iteration_reset(s, p);
  // build_emitter_path, :379 (mode != PathTracing)
  if (o.mode != ETX_BDPT_MODE_PATH_TRACING) {
    {
      ScopedTimer t(ctx, kTimerGenerate);
      launch_bdpt_light_generate(s, p, it);
    }
    const bool to_camera = o.connect_to_camera != 0;
    rc = run_bounce_loop(
      ctx,
      [&](uint32_t set, uint32_t max_items) {
        {
          ScopedTimer t(ctx, kTimerShadeLight);
          launch_bdpt_light_shade(s, p, it, set, max_items, simple);
          if (ctx->scene.has_subsurface)
            launch_bdpt_walk(s, p, it, false, set, max_items, simple);
          if (to_camera)
            launch_bdpt_connect_camera(s, p, it, max_items, simple);
        }
        if (to_camera) {
          ScopedTimer t(ctx, kTimerTraceShadow);
          launch_trace_shadow(s, p, uint32_t(std::min<uint64_t>(uint64_t(max_items) * 2ull, p.shadow.capacity)), flat);
        }
      },
      [&](uint32_t, uint32_t) {}, rounds, kStatRaysLight, false);
    if (rc)
      return rc;
  }
  // build_camera_path, :388 (mode != LightTracing)
  if (o.mode != ETX_BDPT_MODE_LIGHT_TRACING) {
    {
      ScopedTimer t(ctx, kTimerGenerate);
      launch_bdpt_camera_generate(s, p, it);
    }
    const bool to_light = o.connect_to_light != 0;
    const bool vertices = (o.connect_vertices != 0) && (o.mode == ETX_BDPT_MODE_FULL);
    rc = run_bounce_loop(
      ctx,
      [&](uint32_t set, uint32_t max_items) {
        {
          ScopedTimer t(ctx, kTimerShadeCamera);
          launch_bdpt_camera_shade(s, p, it, set, max_items, simple);
          if (ctx->scene.has_subsurface)
            launch_bdpt_walk(s, p, it, true, set, max_items, simple);
          if (to_light)
            launch_bdpt_connect_light(s, p, it, max_items, simple);
        }
        if (vertices) {
          ScopedTimer t(ctx, kTimerConnect);
          launch_bdpt_expand_pairs(s, p, it, max_items);
          launch_bdpt_connect_pairs(s, p, it, max_items, simple);
        }
        if (to_light || vertices) {
          ScopedTimer t(ctx, kTimerTraceShadow);
          launch_trace_shadow(s, p, uint32_t(std::min<uint64_t>(uint64_t(max_items) * 6ull, p.shadow.capacity)), flat);
        }
      },
      [&](uint32_t, uint32_t) {}, rounds, kStatRaysCamera, false);
    if (rc)
      return rc;
  }
  {
    CommitSection commit(ctx);
    launch_vcm_commit(s, p.camera_sum, p.light_sum, ctx->pipe.camera_sum, ctx->pipe.light_sum, p.capacity, p.counters);
  }
  launch_stats_finalize(s, p);
  ctx->stats.wavefront_bounces = rounds;
  return 0;
}

void collect_stats(etx_hip_context* ctx) {
  auto& st = ctx->stats;
  double ms[kTimerCount] = {};
  uint64_t launches[kTimerCount] = {};
  for (const auto& span : ctx->spans) {
    float t = 0.0f;
    if (hipEventElapsedTime(&t, span.begin, span.end) == hipSuccess) {
      ms[span.id] += t;
      launches[span.id]++;
    }
  }
  st.ms_trace_closest = ms[kTimerTraceClosest];
  st.launches_trace_closest = launches[kTimerTraceClosest];
  st.ms_shade_light = ms[kTimerShadeLight];
  st.ms_shade_camera = ms[kTimerShadeCamera];
  st.ms_connect = ms[kTimerConnect];
  st.ms_merge = ms[kTimerMerge];
  st.ms_grid_build = ms[kTimerGridBuild];
  st.ms_generate = ms[kTimerGenerate];
  st.ms_trace_shadow = ms[kTimerTraceShadow];
  st.launches_trace_shadow = launches[kTimerTraceShadow];
  ctx->spans.clear();
  ctx->events_used = 0;

  const uint32_t* c = ctx->host_counters;
  auto u64 = [&](uint32_t i) {
    uint64_t v;
    memcpy(&v, c + i, sizeof(v));
    return v;
  };
  st.rays_extension = u64(kStatRaysExtension);
  st.rays_shadow = u64(kStatRaysShadow);
  st.light_vertices = c[kCntLightVertices];
  st.camera_vertices = u64(kStatCameraVertices);
  st.photons_examined = u64(kStatPhotonsExamined);
  st.photons_merged = u64(kStatPhotonsMerged);
  st.splats = u64(kStatSplats);
  st.rays_light = u64(kStatRaysLight);
  st.rays_camera = u64(kStatRaysCamera);
  st.pairs = u64(kStatPairs);
  st.endpoints = u64(kStatEndpoints);
  st.active_pixels = st.last_active_pixels = u64(kStatActivePixels);
  st.boundary_crossings = u64(kStatCrossings);
  st.overflow_flags = c[kCntOverflow];
  st.nonfinite_dropped = c[kCntNonFinite];
#if defined(ETX_HIP_DEBUG_COUNTERS)
  fprintf(stderr, "[dbg] it %u: lv %u cv %llu pairs %llu shadow %llu list_sum %llu list_max %u ext %llu\n", st.current_iteration, c[kCntLightVertices], (unsigned long long)u64(kStatCameraVertices),
    (unsigned long long)u64(kDbgBase + 0), (unsigned long long)st.rays_shadow, (unsigned long long)u64(kDbgBase + 2), c[kDbgBase + 4], (unsigned long long)st.rays_extension);
#endif
}

etx_hip_context* public_context(etx_hip_context* lane) {
  return lane->owner ? lane->owner : lane;
}

// A pool overflowed: double what overflowed (bounded by the record limit and etx_hip_set_pool_policy's byte limit) in the sizes every lane
// adopts. False: nothing can grow any more.
bool grow_pools(etx_hip_context* lane, uint32_t flags) {
  etx_hip_context* pub = public_context(lane);
  std::lock_guard<std::mutex> lock(pub->shared_mutex);
  etx_hip_context::PoolSizes w = pub->pool_wanted;
  // another lane may have grown the pools since this lane adopted its sizes: take those first
  if ((w == lane->pool_sizes) == false)
    return true;
  auto twice = [](uint32_t v, uint64_t limit) { return uint32_t(std::min<uint64_t>(uint64_t(v) * 2ull, limit)); };
  if (flags & kOverflowLightVertices)
    w.light_vertices = twice(w.light_vertices, kPoolRecordLimit);
  if (flags & kOverflowPairs)
    w.pairs = twice(w.pairs, kPoolRecordLimit);
  if (flags & kOverflowCameraVertices)
    w.camera_vertices = twice(w.camera_vertices, kPoolRecordLimit);
  if (flags & kOverflowEndpoints)
    w.endpoints = twice(w.endpoints, kPoolRecordLimit);
  if (flags & kOverflowShadow)
    w.shadow = twice(w.shadow, 0xfffffff0ull);
  w.shadow = std::max(w.shadow, uint32_t(std::min<uint64_t>(uint64_t(w.pairs) + 2ull * w.camera_vertices, 0xfffffff0ull)));
  if (w == pub->pool_wanted)
    return false;
  if ((pub->pool_limit_bytes != 0u) && (pool_bytes_for(w, pub->grid_wanted || lane->grid_wanted) > pub->pool_limit_bytes))
    return false;
  pub->pool_wanted = w;
  pub->totals.pool_grows += 1u;
  return true;
}

// Renders one iteration on `lane` (worker thread of that lane) and folds its statistics into the public context. An iteration whose pools
// overflowed has not been committed to the film (k_vcm_commit): the pools grow and the same iteration is rendered again.
int execute_iteration(etx_hip_context* lane, uint32_t iteration) {
  etx_hip_context* pub = public_context(lane);
  for (uint32_t attempt = 0;; ++attempt) {
    etx_hip_context::PoolSizes wanted;
    {
      std::lock_guard<std::mutex> lock(pub->shared_mutex);
      wanted = pub->pool_wanted;
    }
    if ((wanted == lane->pool_sizes) == false) {
      HIP_OK(lane, hipStreamSynchronize(lane->stream));
      const etx_hip_context::PoolSizes previous = lane->pool_sizes;
      if (allocate_pools(lane, wanted) != 0) {
        // out of device memory while growing: the lane goes back to the pools it had (never left without any), every lane is told to stay
        // there, and the iteration that needed the larger pools fails as an overflow that cannot grow
        const std::string why = lane->error;
        {
          std::lock_guard<std::mutex> lock(pub->shared_mutex);
          if (previous.light_vertices != 0u)
            pub->pool_wanted = previous;
        }
        if ((previous.light_vertices == 0u) || (allocate_pools(lane, previous) != 0))
          return ETX_HIP_ERROR_HIP;  // not even the previous sizes: lane->error says which allocation
        lane->error = "device pools cannot grow for iteration " + std::to_string(iteration) + ": " + why;
        return ETX_HIP_ERROR_OVERFLOW;
      }
    }
    lane->retry_attempt = attempt != 0u;
    HIP_OK(lane, hipEventRecord(lane->iteration_begin, lane->stream));
    lane->stats = {};
    lane->stats.current_iteration = iteration;
    int rc = (lane->integrator == ETX_HIP_INTEGRATOR_PT)     ? render_pt_iteration(lane, iteration)
             : (lane->integrator == ETX_HIP_INTEGRATOR_BDPT) ? render_bdpt_iteration(lane, iteration)
                                                             : render_vcm_iteration(lane, iteration);
    if (rc)
      return rc;
    HIP_OK(lane, hipEventRecord(lane->iteration_end, lane->stream));
    // the bounce loop already synchronised on the counters; the tail (nothing after the last read) is short
    HIP_OK(lane, hipEventSynchronize(lane->iteration_end));
    rc = read_counters(lane);
    if (rc)
      return rc;
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, lane->iteration_begin, lane->iteration_end);
    collect_stats(lane);
    lane->stats.last_iteration_time = double(ms) * 1.0e-3;
    // the next iteration's launch plans: what every round of this one found (the stream is idle, every mirror entry has arrived)
    for (etx_hip_context::PassPlan& plan : lane->plans) {
      plan.entering.clear();
      for (uint32_t r = 0; (r < plan.rounds_enqueued) && (r < kRoundMirrorSlots); ++r) {
        const uint32_t tag = plan.first_tag + r;
        const unsigned long long entry = lane->round_mirror[tag & (kRoundMirrorSlots - 1u)];
        if (uint32_t(entry >> 32u) != tag + 1u)
          break;  // overwritten by a later round of a pass longer than the ring: no plan, the next iteration polls
        plan.entering.push_back(uint32_t(entry & 0xffffffffull));
      }
      if (plan.entering.size() != plan.rounds_enqueued)
        plan.entering.clear();
      plan.rounds_enqueued = 0u;
    }
    // A scheduled pass enqueues only the rounds its plan holds and the next plan is read back from exactly those: a plan can shrink (noise
    // pushes the last round under the tail threshold) but never grow again, and the tail kernel would absorb that round for good (ADVICE
    // round 4). Every kPlanLifetime-th iteration of a lane therefore polls and rebuilds its plans from what the device reports.
    constexpr uint32_t kPlanLifetime = 32u;
    if (++lane->iterations_on_plan >= kPlanLifetime) {
      lane->iterations_on_plan = 0u;
      for (etx_hip_context::PassPlan& plan : lane->plans)
        plan.entering.clear();
    }
    const uint32_t flags = lane->stats.overflow_flags;
    if (flags == 0u)
      return ETX_HIP_OK;
    // the path tracer commits unconditionally (its queues are bounded by construction: two requests per path and round); a traversal stack
    // cannot grow
    const bool growable = (lane->integrator != ETX_HIP_INTEGRATOR_PT) && ((flags & kOverflowStack) == 0u) && (attempt < 8u);
    if ((growable == false) || (grow_pools(lane, flags) == false)) {
      lane->error = "device pool overflow in iteration " + std::to_string(iteration) + " (flags " + std::to_string(flags) +
                    "): light vertex pool (1) / traversal stack (2) / connection pair buffer (4) / shadow queue (8) / camera vertex pool (16) / endpoint queue (32) "
                    "cannot grow any further (etx_hip_set_pool_policy limits a lane's pools to " + std::to_string(pub->pool_limit_bytes) + " bytes; 0 = the device's memory)";
      return ETX_HIP_ERROR_OVERFLOW;
    }
  }
}

void lane_worker(etx_hip_context* lane) {
  (void)hipSetDevice(lane->device);
  etx_hip_context* pub = public_context(lane);
  for (;;) {
    uint32_t iteration = 0;
    {
      std::unique_lock<std::mutex> lock(lane->lane_mutex);
      lane->lane_cv.wait(lock, [&] { return lane->quit || (lane->jobs.empty() == false); });
      if (lane->quit && lane->jobs.empty())
        return;
      iteration = lane->jobs.front();
      lane->jobs.pop_front();
    }
    int rc = execute_iteration(lane, iteration);
    {
      std::lock_guard<std::mutex> lock(pub->shared_mutex);
      auto& t = pub->totals;
      const auto& s = lane->stats;
      if (rc == ETX_HIP_OK) {
        t.completed_iterations += 1;
        t.last_iteration_time = s.last_iteration_time;
        t.rays_extension += s.rays_extension, t.rays_shadow += s.rays_shadow;
        t.light_vertices += s.light_vertices, t.camera_vertices += s.camera_vertices;
        t.photons_examined += s.photons_examined, t.photons_merged += s.photons_merged, t.splats += s.splats;
        t.rays_light += s.rays_light, t.rays_camera += s.rays_camera, t.pairs += s.pairs, t.endpoints += s.endpoints;
        t.active_pixels += s.active_pixels, t.last_active_pixels = s.last_active_pixels;
        t.boundary_crossings += s.boundary_crossings;
        t.wavefront_bounces += s.wavefront_bounces;
        t.ms_trace_closest += s.ms_trace_closest, t.ms_trace_shadow += s.ms_trace_shadow;
        t.ms_shade_light += s.ms_shade_light, t.ms_shade_camera += s.ms_shade_camera;
        t.ms_connect += s.ms_connect, t.ms_merge += s.ms_merge, t.ms_grid_build += s.ms_grid_build, t.ms_generate += s.ms_generate;
        t.launches_trace_closest += s.launches_trace_closest, t.launches_trace_shadow += s.launches_trace_shadow;
        pub->local_iterations += 1;
      } else if (pub->sticky_error == 0) {
        pub->sticky_error = rc;
        pub->sticky_error_text = lane->error;
      }
      t.overflow_flags |= s.overflow_flags;
      t.nonfinite_dropped += s.nonfinite_dropped;
      lane->lane_busy = false;
      pub->jobs_in_flight -= 1;
      if (pub->jobs_in_flight == 0u)
        t.total_time += std::chrono::duration<double>(std::chrono::steady_clock::now() - pub->busy_since).count();
    }
    pub->idle_cv.notify_all();
  }
}

// Waits until no iteration is queued or running; returns the sticky error of a failed iteration, if any.
int wait_idle(etx_hip_context* pub) {
  std::unique_lock<std::mutex> lock(pub->shared_mutex);
  pub->idle_cv.wait(lock, [&] { return pub->jobs_in_flight == 0u; });
  if (pub->sticky_error) {
    pub->error = pub->sticky_error_text;
    return pub->sticky_error;
  }
  return ETX_HIP_OK;
}

int init_lane(etx_hip_context* lane, int device, std::string& error) {
  lane->device = device;
  if (hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking) != hipSuccess) {
    error = "hipStreamCreate failed";
    return ETX_HIP_ERROR_HIP;
  }
  if ((hipEventCreate(&lane->iteration_begin) != hipSuccess) || (hipEventCreate(&lane->iteration_end) != hipSuccess) ||
      (hipEventCreateWithFlags(&lane->commit_done, hipEventDisableTiming) != hipSuccess)) {
    error = "hipEventCreate failed";
    return ETX_HIP_ERROR_HIP;
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&lane->host_counters), kCounterCount * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
    error = "hipHostMalloc failed";
    return ETX_HIP_ERROR_HIP;
  }
  memset(lane->host_counters, 0, kCounterCount * sizeof(uint32_t));
  if (hipHostMalloc(reinterpret_cast<void**>(&lane->round_mirror), kRoundMirrorSlots * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
    error = "hipHostMalloc failed";
    return ETX_HIP_ERROR_HIP;
  }
  memset(lane->round_mirror, 0, kRoundMirrorSlots * sizeof(unsigned long long));
  lane->check_interval = std::max(1u, etxh::tuning_knob("ETX_HIP_CHECK_INTERVAL", lane->check_interval));
  lane->tail_divisor = etxh::tuning_knob("ETX_HIP_TAIL_DIVISOR", lane->tail_divisor);
  lane->scheduled_passes = etxh::tuning_knob("ETX_HIP_SCHEDULED_PASSES", 1u) != 0u;
  lane->timer_mask = etxh::tuning_knob("ETX_HIP_TIMERS", lane->timer_mask);
  lane->debug_flags = etxh::tuning_knob("ETX_HIP_DEBUG_FLAGS", 0u);
  lane->worker = std::thread(lane_worker, lane);
  return ETX_HIP_OK;
}

void stop_lane_worker(etx_hip_context* lane) {
  if (lane->worker.joinable()) {
    {
      std::lock_guard<std::mutex> lock(lane->lane_mutex);
      lane->quit = true;
    }
    lane->lane_cv.notify_all();
    lane->worker.join();
  }
}

// Helper lanes (thread, stream, events, pinned mirrors) up to `wanted` lanes in all; their pools follow in etx_hip_upload_scene / etx_hip_begin.
int ensure_lanes(etx_hip_context* context, uint32_t wanted, std::string& error) {
  while (context->helpers.size() + 1u < wanted) {
    auto* helper = new etx_hip_context();
    helper->owner = context;
    context->helpers.push_back(helper);
    if (int rc = init_lane(helper, context->device, error))
      return rc;
    // what etx_hip_set_timers / etx_hip_set_debug_flags told the lanes that existed then
    helper->timer_mask = context->timer_mask;
    helper->debug_flags = helper->pipe.debug_flags = context->debug_flags;
  }
  return ETX_HIP_OK;
}

void destroy_lane(etx_hip_context* lane) {
  stop_lane_worker(lane);
  if (lane->stream)
    (void)hipStreamSynchronize(lane->stream);
  release_pipeline(lane);
  lane->scene.release();
  for (hipEvent_t e : lane->event_pool)
    (void)hipEventDestroy(e);
  if (lane->iteration_begin)
    (void)hipEventDestroy(lane->iteration_begin);
  if (lane->iteration_end)
    (void)hipEventDestroy(lane->iteration_end);
  if (lane->commit_done)
    (void)hipEventDestroy(lane->commit_done);
  if (lane->host_counters)
    (void)hipHostFree(lane->host_counters);
  if (lane->round_mirror)
    (void)hipHostFree(lane->round_mirror);
  if (lane->stream)
    (void)hipStreamDestroy(lane->stream);
}

}  // namespace

extern "C" {

void etx_hip_destroy(etx_hip_context* context);

int etx_hip_abi_version(void) {
  return ETX_HIP_ABI_VERSION;
}

const char* etx_hip_last_error(const etx_hip_context* context) {
  return context ? context->error.c_str() : g_create_error.c_str();
}

void etx_hip_internal_rccl_versions(int* mapped, int* built);

int etx_hip_runtime_info(int out_versions[4]) {
  if (out_versions == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  out_versions[0] = out_versions[2] = 0;
  out_versions[1] = HIP_VERSION;
  (void)hipRuntimeGetVersion(&out_versions[0]);
  etx_hip_internal_rccl_versions(&out_versions[2], &out_versions[3]);
  return ETX_HIP_OK;
}

namespace {
// The runtime the dynamic loader bound this library to. A package that bundles its own ROCm under the same sonames (a PyTorch wheel: libamdhip64.so.7,
// libhsa-runtime64.so.1, librccl.so.1 of ROCm 7.0.2 in torch/lib) and is imported BEFORE libetx_hip.so makes the loader resolve this library's
// NEEDED entries to those copies: code objects and host stubs built by a newer hipcc then run on an older runtime. Round 5's GPU suite ran like
// that and crashed in one of six processes; the library refuses the configuration instead of running on it.
int check_runtime(std::string& error) {
  int versions[4] = {};
  (void)etx_hip_runtime_info(versions);
  const bool verbose = getenv("ETX_HIP_VERBOSE") != nullptr;
  std::string mapped;
  if (FILE* maps = fopen("/proc/self/maps", "r")) {
    char line[1024];
    std::string last;
    while (fgets(line, sizeof(line), maps)) {
      const char* path = strchr(line, '/');
      if ((path == nullptr) || ((strstr(path, "libamdhip64") == nullptr) && (strstr(path, "libhsa-runtime64") == nullptr) && (strstr(path, "librccl") == nullptr)))
        continue;
      std::string p(path);
      while ((p.empty() == false) && ((p.back() == '\n') || (p.back() == ' ')))
        p.pop_back();
      if (p != last)
        mapped += (mapped.empty() ? "" : ", ") + p;
      last = p;
    }
    fclose(maps);
  }
  if (verbose)
    fprintf(stderr, "[etx_hip] HIP runtime %d (built against %d), RCCL %d (built against %d); mapped: %s\n", versions[0], versions[1], versions[2], versions[3], mapped.c_str());
  if ((versions[0] != 0) && (versions[0] / 100000 < versions[1] / 100000)) {
    const char* allow = getenv("ETX_HIP_ALLOW_OLDER_RUNTIME");
    if ((allow == nullptr) || (allow[0] == '0')) {
      error = "the HIP runtime mapped into this process is version " + std::to_string(versions[0]) + ", older than the " + std::to_string(versions[1]) +
              " libetx_hip.so was built against (mapped: " + mapped + "). Something that bundles its own ROCm runtime (a PyTorch wheel) was loaded first: load libetx_hip.so "
              "before it, or set ETX_HIP_ALLOW_OLDER_RUNTIME=1 to run on the older runtime anyway";
      return ETX_HIP_ERROR_HIP;
    }
  }
  return ETX_HIP_OK;
}
}  // namespace

int etx_hip_create(int device, etx_hip_context** out_context) {
  if (out_context == nullptr) {
    g_create_error = "out_context is null";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  *out_context = nullptr;
  if (int rc = check_runtime(g_create_error))
    return rc;
  int count = 0;
  if ((hipGetDeviceCount(&count) != hipSuccess) || (count == 0)) {
    g_create_error = "no HIP device available (this backend has no CPU path)";
    return ETX_HIP_ERROR_NO_DEVICE;
  }
  if ((device < 0) || (device >= count)) {
    g_create_error = "device index out of range";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  hipDeviceProp_t props = {};
  if (hipGetDeviceProperties(&props, device) != hipSuccess) {
    g_create_error = "hipGetDeviceProperties failed";
    return ETX_HIP_ERROR_HIP;
  }
  if (strncmp(props.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is ") + props.gcnArchName + ", this library is built for gfx950 only";
    return ETX_HIP_ERROR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_create_error = "hipSetDevice failed";
    return ETX_HIP_ERROR_HIP;
  }
  auto ctx = std::make_unique<etx_hip_context>();
  ctx->scene.transfer = &ctx->transfer;
  int rc = init_lane(ctx.get(), device, g_create_error);
  // ETX_HIP_LANES: iterations in flight. Measured at 1080p (fog Cornell): 1 lane 55, 2 lanes 78, 3 lanes 89, 4 lanes 92,
  // 6 lanes 93 Msamples/s - the thin tails and small bounces of one iteration hide behind the wide bounces of the others.
  // The bidirectional integrator gains from more (configs[3]: 25.3 -> 26.6 with six, configs[4]: 192 -> 201): its walk / shadow / connect
  // kernels are latency-bound and narrow. Lanes past the fourth get their pools on the first etx_hip_begin that uses them
  // (lanes_for_integrator), so VCM and path-tracing renders do not pay their memory.
  int lanes = kBidirectionalLanes;
  ctx->base_lanes = kBaseLanes;
  if (const char* e = getenv("ETX_HIP_LANES"))
    ctx->base_lanes = lanes = std::min(8, std::max(1, atoi(e)));
  ctx->active_lanes = ctx->base_lanes;
  ctx->max_lanes = uint32_t(lanes);
  if (rc == ETX_HIP_OK)
    rc = ensure_lanes(ctx.get(), ctx->base_lanes, g_create_error);  // the bidirectional integrator's extra lanes: at its first etx_hip_begin
  if (rc != ETX_HIP_OK) {
    etx_hip_destroy(ctx.release());
    return rc;
  }
  *out_context = ctx.release();
  return ETX_HIP_OK;
}

void etx_hip_comm_destroy_internal(etx_hip_context* context);

void etx_hip_destroy(etx_hip_context* context) {
  if (context == nullptr)
    return;
  (void)hipSetDevice(context->device);
  // iterations still in flight finish first: their commits use the reduce state (CommitSection) that etx_hip_comm_destroy_internal tears down,
  // and every lane's worker - the public lane's too - is joined before that
  const char* legacy = getenv("ETX_HIP_DEBUG_LEGACY");  // DIAGNOSTIC (round 6): bit 1 = the round-5 teardown order
  if ((legacy != nullptr) && ((atoi(legacy) & 2) != 0)) {
    for (etx_hip_context* helper : context->helpers) {
      destroy_lane(helper);
      delete helper;
    }
    context->helpers.clear();
  } else {
    (void)wait_idle(context);
    for (etx_hip_context* helper : context->helpers)
      stop_lane_worker(helper);
    stop_lane_worker(context);
  }
  etx_hip_comm_destroy_internal(context);
  for (etx_hip_context* helper : context->helpers) {
    destroy_lane(helper);
    delete helper;
  }
  context->helpers.clear();
  destroy_lane(context);
  for (uint8_t* table : context->bluenoise)
    if (table)
      (void)hipFree(table);
  if (context->cie_table)
    (void)hipFree(context->cie_table);
  if (context->rgb_response_table)
    (void)hipFree(context->rgb_response_table);
  if (context->read_resolve)
    (void)hipFree(context->read_resolve);
  if (context->read_staging)
    (void)hipHostFree(context->read_staging);
  if (context->read_event)
    (void)hipEventDestroy(context->read_event);
  if (context->read_stream)
    (void)hipStreamDestroy(context->read_stream);
  context->transfer.release();
  delete context;
}

int etx_hip_upload_scene(etx_hip_context* context, const etx_abi_scene* scene, const etx_abi_camera* camera) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  HIP_OK(context, hipSetDevice(context->device));
  (void)wait_idle(context);
  HIP_OK(context, hipStreamSynchronize(context->stream));
  context->scene_ready = false;
  context->armed = false;
  if (context->read_pending) {
    (void)hipEventSynchronize(context->read_event);
    context->read_pending = false;
  }
  // a reduce in flight reads the film sums allocate_pipeline is about to free, and the reduced copy is the OLD scene's film:
  // wait for it, forget it; the buffers follow the new film size below
  if (int reset_rc = etx_hip_internal_reduce_reset(context))
    return reset_rc;
  for (etx_hip_context* helper : context->helpers)
    release_pipeline(helper);
  int rc = etxh::build_device_scene(scene, camera, context->scene, context->error);
  if (rc)
    return rc;
  context->pool_wanted = {};  // the pools start over at what this scene typically needs (initial_pool_sizes)
  rc = allocate_pipeline(context);
  if (rc)
    return rc;
  if ((rc = allocate_base_lanes(context)))
    return rc;
  HIP_OK(context, hipDeviceSynchronize());
  context->scene_ready = true;
  if (context->comm != nullptr) {  // the reduce buffers at the new film size: a rank finds out here, not inside a collective, that it cannot take part
    if ((rc = etx_hip_internal_reduce_allocate(context)))
      return rc;
  }
  return ETX_HIP_OK;
}

int etx_hip_set_bvh_builder(etx_hip_context* context, int builder) {
  if ((context == nullptr) || ((builder != ETX_HIP_BVH_HOST_SAH) && (builder != ETX_HIP_BVH_DEVICE_LBVH)))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  context->scene.device_bvh_build = builder == ETX_HIP_BVH_DEVICE_LBVH;
  return ETX_HIP_OK;
}

int etx_hip_bvh_info(etx_hip_context* context, uint32_t out_info[4], double* out_build_ms) {
  if ((context == nullptr) || (out_info == nullptr))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->scene_ready == false) {
    context->error = "etx_hip_bvh_info: no scene uploaded";
    return ETX_HIP_ERROR_STATE;
  }
  const auto& d = context->scene.host_copy;
  out_info[0] = d.bvh_node_count, out_info[1] = d.bvh_tri_count, out_info[2] = d.bvh_depth | (d.bvh_stack_need << 16u), out_info[3] = uint32_t(std::min<size_t>(context->scene.bvh_bytes, 0xffffffffu));
  if (out_build_ms != nullptr)
    *out_build_ms = context->scene.bvh_build_ms;
  return ETX_HIP_OK;
}

int etx_hip_update_scene(etx_hip_context* context, const etx_abi_scene* scene, const etx_abi_camera* camera, uint32_t changed) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->scene_ready == false) {
    context->error = "etx_hip_update_scene: no scene uploaded (etx_hip_upload_scene first)";
    return ETX_HIP_ERROR_STATE;
  }
  if (((changed & ~uint32_t(ETX_HIP_CHANGED_CAMERA | ETX_HIP_CHANGED_MATERIALS | ETX_HIP_CHANGED_POSITIONS | ETX_HIP_REBUILD_BVH)) != 0u) ||
      ((changed & ETX_HIP_REBUILD_BVH) && ((changed & ETX_HIP_CHANGED_POSITIONS) == 0u))) {
    context->error = "etx_hip_update_scene: unknown bits in `changed` (anything else that changed needs etx_hip_upload_scene; ETX_HIP_REBUILD_BVH goes with ETX_HIP_CHANGED_POSITIONS)";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if ((camera != nullptr) && ((camera->film_size.x != context->scene.film_w) || (camera->film_size.y != context->scene.film_h))) {
    context->error = "etx_hip_update_scene: the film size changed; use etx_hip_upload_scene";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  HIP_OK(context, hipSetDevice(context->device));
  (void)wait_idle(context);
  HIP_OK(context, hipStreamSynchronize(context->stream));
  context->scene_ready = false;  // a failure below leaves no scene, like a failed upload
  context->armed = false;
  if (context->read_pending) {
    (void)hipEventSynchronize(context->read_event);
    context->read_pending = false;
  }
  if (int reset_rc = etx_hip_internal_reduce_reset(context))  // as etx_hip_upload_scene: the film sums are reallocated below
    return reset_rc;
  // the tables (materials, spectra, emitters, media parameters, scene scalars, camera) are small and always rebuilt; vertices,
  // triangles, BVH, image pixels and density grids stay on the device
  int rc = etxh::build_device_scene(scene, camera, context->scene, context->error, /* keep geometry and images */ true);
  if (rc)
    return rc;
  if ((changed & (ETX_HIP_CHANGED_POSITIONS | ETX_HIP_CHANGED_MATERIALS)) &&
      (rc = etxh::update_device_geometry(scene, context->scene, context->stream, (changed & ETX_HIP_CHANGED_POSITIONS) != 0u, (changed & ETX_HIP_REBUILD_BVH) != 0u, context->error)))
    return rc;
  // pool sizes follow the materials in use (subsurface scenes keep more vertices per path): the lanes' pipelines are set up again
  for (etx_hip_context* helper : context->helpers)
    release_pipeline(helper);
  context->pool_wanted = {};
  rc = allocate_pipeline(context);
  if (rc)
    return rc;
  if ((rc = allocate_base_lanes(context)))
    return rc;
  HIP_OK(context, hipDeviceSynchronize());
  context->scene_ready = true;
  return ETX_HIP_OK;
}

uint32_t etx_hip_lanes(const etx_hip_context* context, int integrator) {
  return (context == nullptr) ? 0u : lanes_for_integrator(context, integrator);
}

int etx_hip_set_pool_policy(etx_hip_context* context, uint32_t initial_light_vertices_per_path, size_t max_pool_bytes_per_lane) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  context->pool_initial_per_path = initial_light_vertices_per_path;
  context->pool_limit_bytes = max_pool_bytes_per_lane;
  return ETX_HIP_OK;
}

int etx_hip_upload_bluenoise(etx_hip_context* context, uint32_t set_index, const uint8_t* values, size_t bytes) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  constexpr size_t kTableBytes = size_t(128) * 128 * 256 * 8;
  if ((set_index >= kBlueNoiseSets) || (values == nullptr) || (bytes != kTableBytes)) {
    context->error = "etx_hip_upload_bluenoise: set_index must be 0..8 and the table 128*128*256*8 bytes";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  HIP_OK(context, hipSetDevice(context->device));
  if (context->bluenoise[set_index] == nullptr)
    HIP_OK(context, hipMalloc(reinterpret_cast<void**>(&context->bluenoise[set_index]), kTableBytes));
  TRANSFER_OK(context, context->transfer.to_device(context->bluenoise[set_index], values, kTableBytes, nullptr, context->error));
  return ETX_HIP_OK;
}

namespace {
// BNSampler's class for scene.samples (thirdparty/bluenoise/bluenoise.cxx:78-92)
int select_bluenoise(etx_hip_context* context, const char* option_name) {
  uint32_t samples = context->scene.host_copy.samples;
  samples = (samples == 0u) ? 1u : std::min(samples, 256u);
  const uint32_t set_index = 31u - uint32_t(__builtin_clz(next_pow2(samples)));
  if (context->bluenoise[set_index] == nullptr) {
    context->error = std::string("options.blue_noise: the blue-noise samples of class ") + std::to_string(1u << set_index) + " spp (set " + std::to_string(set_index) +
                     ") have not been uploaded (etx_hip_upload_bluenoise); upload them or set " + option_name + "=false";
    return ETX_HIP_ERROR_UNSUPPORTED;
  }
  context->active_bluenoise = context->bluenoise[set_index];
  return ETX_HIP_OK;
}
}  // namespace

int etx_hip_upload_cie_table(etx_hip_context* context, const float* xyz, uint32_t count, float first_wavelength) {
  if ((context == nullptr) || (xyz == nullptr) || (count < 2u) || (count > 4096u))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  (void)wait_idle(context);
  HIP_OK(context, hipSetDevice(context->device));
  std::vector<float4> table(count);
  double y_integral = 0.0;
  for (uint32_t i = 0; i < count; ++i) {
    table[i] = make_float4(xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2], 0.0f);
    y_integral += double(xyz[3 * i + 1]);
  }
  // kYIntegral() accumulates in float (spectrum.hxx:188-194)
  float y_sum = 0.0f;
  for (uint32_t i = 0; i < count; ++i)
    y_sum += xyz[3 * i + 1];
  if (context->cie_table)
    (void)hipFree(context->cie_table);
  context->cie_table = nullptr;
  HIP_OK(context, hipMalloc(reinterpret_cast<void**>(&context->cie_table), count * sizeof(float4)));
  TRANSFER_OK(context, context->transfer.to_device(context->cie_table, table.data(), count * sizeof(float4), nullptr, context->error));
  context->cie_count = count;
  context->cie_first = first_wavelength;
  context->cie_y_scale = (y_sum > 0.0f) ? 1.0f / y_sum : 0.0f;
  (void)y_integral;
  return ETX_HIP_OK;
}

int etx_hip_upload_rgb_response(etx_hip_context* context, const float* rgb, uint32_t count, float first_wavelength) {
  if ((context == nullptr) || (rgb == nullptr) || (count < 2u) || (count > 4096u))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  (void)wait_idle(context);
  HIP_OK(context, hipSetDevice(context->device));
  std::vector<float4> table(count);
  for (uint32_t i = 0; i < count; ++i)
    table[i] = make_float4(rgb[3 * i + 0], rgb[3 * i + 1], rgb[3 * i + 2], 0.0f);
  if (context->rgb_response_table)
    (void)hipFree(context->rgb_response_table);
  context->rgb_response_table = nullptr;
  HIP_OK(context, hipMalloc(reinterpret_cast<void**>(&context->rgb_response_table), count * sizeof(float4)));
  TRANSFER_OK(context, context->transfer.to_device(context->rgb_response_table, table.data(), count * sizeof(float4), nullptr, context->error));
  context->rgb_response_count = count;
  context->rgb_response_first = first_wavelength;
  return ETX_HIP_OK;
}

int etx_hip_begin(etx_hip_context* context, int integrator, const void* options, size_t options_size, uint32_t first_iteration, uint32_t iteration_stride) {
  return etx_hip_begin_ex(context, integrator, options, options_size, first_iteration, iteration_stride, 0u, 1u);
}

int etx_hip_begin_ex(etx_hip_context* context, int integrator, const void* options, size_t options_size, uint32_t first_iteration, uint32_t iteration_stride, uint32_t pixel_first,
  uint32_t pixel_stride) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  float noise_threshold = 0.0f;
  if (context->scene_ready == false) {
    context->error = "etx_hip_begin: no scene uploaded";
    return ETX_HIP_ERROR_STATE;
  }
  if (iteration_stride == 0) {
    context->error = "iteration_stride must be >= 1";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if ((pixel_stride == 0u) || (pixel_first >= pixel_stride)) {
    context->error = "pixel sharding: pixel_stride must be >= 1 and pixel_first below it";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if ((integrator == ETX_HIP_INTEGRATOR_VCM) && (pixel_stride != 1u)) {
    context->error = "VCM cannot be sharded by pixels: the photon map of an iteration holds the light paths of EVERY pixel and the merge is normalised by their number "
                     "(vcm_cpu.cxx:100-113); shard its iterations (first_iteration / iteration_stride)";
    return ETX_HIP_ERROR_UNSUPPORTED;
  }
  (void)wait_idle(context);  // iterations of the previous run
  const uint32_t lanes_wanted = lanes_for_integrator(context, integrator);
  if (integrator == ETX_HIP_INTEGRATOR_VCM) {
    if ((options == nullptr) || (options_size != sizeof(etx_abi_vcm_options))) {
      context->error = "VCM expects etx_abi_vcm_options (32 bytes)";
      return ETX_HIP_ERROR_INVALID_ARGUMENT;
    }
    memcpy(&context->vcm_options, options, sizeof(etx_abi_vcm_options));
    context->active_bluenoise = nullptr;
    if (context->vcm_options.blue_noise) {
      int rc = select_bluenoise(context, "vcm-blue_noise");
      if (rc)
        return rc;
    }
    if (context->vcm_options.radius_decay == 0) {
      context->error = "radius_decay must be >= 1";
      return ETX_HIP_ERROR_INVALID_ARGUMENT;
    }
  } else if (integrator == ETX_HIP_INTEGRATOR_PT) {
    if ((options == nullptr) || (options_size != sizeof(etx_abi_pt_options))) {
      context->error = "PT expects etx_abi_pt_options (16 bytes)";
      return ETX_HIP_ERROR_INVALID_ARGUMENT;
    }
    memcpy(&context->pt_options, options, sizeof(etx_abi_pt_options));
    context->active_bluenoise = nullptr;
    if (context->pt_options.blue_noise) {
      int rc = select_bluenoise(context, "bn");
      if (rc)
        return rc;
    }
    // adaptive sampling: CPUPathTracing is the integrator that calls Film::estimate_noise_levels (path_tracing.cxx:99); CPUVCM and
    // CPUBidirectional never do, every pixel stays active for them whatever the threshold
    noise_threshold = context->scene.noise_threshold;
    // An iteration-sharded context (multi-GPU) holds a film of ITS iterations only, and the convergence mask is a property of the whole
    // film: such a run samples every pixel in every iteration (a superset of what the adaptive render samples; the reference's default
    // threshold is 0.1, so this is what every unedited scene gets on several GPUs)
    // - and the mask looks at a pixel's row and column neighbours (film.cxx:283-321), which a pixel-sharded context does not render
    if ((iteration_stride != 1u) || (pixel_stride != 1u))
      noise_threshold = 0.0f;
  } else if (integrator == ETX_HIP_INTEGRATOR_BDPT) {
    if ((options == nullptr) || (options_size != sizeof(etx_abi_bdpt_options))) {
      context->error = "BDPT expects etx_abi_bdpt_options (16 bytes)";
      return ETX_HIP_ERROR_INVALID_ARGUMENT;
    }
    memcpy(&context->bdpt_options, options, sizeof(etx_abi_bdpt_options));
    const uint32_t mode = context->bdpt_options.mode;
    if (mode > ETX_BDPT_MODE_FULL) {
      context->error = "bdpt-mode: unknown value " + std::to_string(mode);
      return ETX_HIP_ERROR_INVALID_ARGUMENT;
    }
    if (context->scene.has_subsurface && (context->scene.sss_media_complete == false)) {
      context->error = "bidirectional integrator: a subsurface material without an interior medium derives its walk medium from its colour and distances, and this one names a spectrum "
                       "outside the scene's table (or the scene's subsurface SCATTER material is textured: the entry vertex' medium instance, bidirectional.cxx:629-633, is a table entry here)";
      return ETX_HIP_ERROR_UNSUPPORTED;
    }
    context->active_bluenoise = nullptr;
    if (context->bdpt_options.blue_noise) {
      int rc = select_bluenoise(context, "bdpt-blue_noise");
      if (rc)
        return rc;
    }
  } else {
    context->error = "integrator " + std::to_string(integrator) + " is not implemented by the device path";
    return ETX_HIP_ERROR_UNSUPPORTED;
  }
  if (context->scene.host_copy.spectral) {
    if (context->cie_table == nullptr) {
      context->error = "spectral scene: the CIE observer table has not been uploaded (etx_hip_upload_cie_table)";
      return ETX_HIP_ERROR_UNSUPPORTED;
    }
    if (context->scene.needs_rgb_response && (context->rgb_response_table == nullptr)) {
      context->error = "spectral scene with RGB images: the rgb_response table has not been uploaded (etx_hip_upload_rgb_response)";
      return ETX_HIP_ERROR_UNSUPPORTED;
    }
    auto patch = [&](etx_hip_context* lane) {
      for (DScene* d : {&lane->scene.host_copy, &lane->pipe.scene}) {
        d->rgb_response = context->rgb_response_table;
        d->rgb_response_count = context->rgb_response_count;
        d->rgb_response_first = context->rgb_response_first;
        d->cie_xyz = context->cie_table;
        d->cie_count = context->cie_count;
        d->cie_first = context->cie_first;
        d->cie_y_scale = context->cie_y_scale;
      }
    };
    patch(context);
    for (etx_hip_context* helper : context->helpers)
      patch(helper);
    HIP_OK(context, hipSetDevice(context->device));
    if (int rc = context->scene.sync_device_copy(context->error))  // the out-of-line BSDF code reads the device-resident header
      return rc;
  }
  HIP_OK(context, hipSetDevice(context->device));
  context->pixel_first = pixel_first, context->pixel_stride = pixel_stride;
  {
    // the growable pools follow the paths a sub pass holds: a context that renders every second pixel starts (and stays, unless it overflows)
    // at half the vertex, pair and queue records; the lanes adopt the size before their next iteration (execute_iteration)
    const uint32_t paths = shard_paths(context->scene.film_w * context->scene.film_h, pixel_first, pixel_stride);
    std::lock_guard<std::mutex> lock(context->shared_mutex);
    if ((paths != context->pool_paths) && (context->pool_paths != 0u)) {
      // another share of the pixels than the pools were sized for: what the pools have GROWN to is kept in proportion (a scene that needed
      // twice the start size needs it on any share), never below the start size of the new share (starting over made the
      // first iterations after every switch overflow and render twice)
      const etx_hip_context::PoolSizes start = initial_pool_sizes(context, std::max(paths, 1u));
      const double ratio = double(std::max(paths, 1u)) / double(context->pool_paths);
      auto scaled = [ratio](uint32_t grown, uint32_t at_least, uint64_t limit) {
        return uint32_t(std::min<uint64_t>(std::max<uint64_t>(uint64_t(std::ceil(double(grown) * ratio)), at_least), limit));
      };
      etx_hip_context::PoolSizes w = context->pool_wanted;
      w.light_vertices = scaled(w.light_vertices, start.light_vertices, kPoolRecordLimit);
      w.camera_vertices = scaled(w.camera_vertices, start.camera_vertices, kPoolRecordLimit);
      w.pairs = scaled(w.pairs, start.pairs, kPoolRecordLimit);
      w.shadow = scaled(w.shadow, start.shadow, 0xfffffff0ull);
      w.endpoints = scaled(w.endpoints, start.endpoints, kPoolRecordLimit);
      context->pool_wanted = w;
      context->pool_paths = paths;
    }
    // etx_hip_set_pool_policy's byte limit bounds the sizes a run STARTS with as well (a large per-path start, the photon grid a VCM run adds)
    const bool with_grid = context->grid_wanted || (integrator == ETX_HIP_INTEGRATOR_VCM);
    if ((context->pool_limit_bytes != 0u) && (pool_bytes_for(context->pool_wanted, with_grid) > context->pool_limit_bytes)) {
      context->error = "etx_hip_begin: the pools this run starts with (" + std::to_string(pool_bytes_for(context->pool_wanted, with_grid)) +
                       " bytes per lane) exceed etx_hip_set_pool_policy's limit of " + std::to_string(context->pool_limit_bytes) + " bytes";
      return ETX_HIP_ERROR_OVERFLOW;
    }
  }
  // the lanes this integrator uses beyond the base ones are created and get their pools now - after every check above, so a refused begin
  // leaves the working set as it was
  if (int rc = ensure_lanes(context, lanes_wanted, context->error))
    return rc;
  for (uint32_t i = 0; i + 1u < lanes_wanted; ++i) {
    etx_hip_context* helper = context->helpers[i];
    if (lane_ready(helper))
      continue;
    helper->scene.borrow(context->scene);
    if (int rc = allocate_pipeline(helper)) {
      context->error = helper->error;
      release_pipeline(helper);
      return rc;
    }
  }
  context->integrator = integrator;
  context->first_iteration = first_iteration;
  context->iteration_stride = iteration_stride;
  context->next_iteration = first_iteration;
  context->local_iterations = 0;
  context->global_iterations = 0;
  if (int rc = etx_hip_internal_reduce_reset(context))  // a reduce of the previous run still in flight is waited for; its result belongs to that run
    return rc;
  context->stats = {};
  {
    std::lock_guard<std::mutex> lock(context->shared_mutex);
    context->totals = {};
    context->sticky_error = 0;
    context->sticky_error_text.clear();
  }
  if (integrator == ETX_HIP_INTEGRATOR_VCM) {
    if (int rc = allocate_photon_grid(context))
      return rc;
    for (uint32_t i = 0; i + 1u < lanes_wanted; ++i) {  // the lanes VCM schedules, not the bidirectional integrator's extra ones
      etx_hip_context* helper = context->helpers[i];
      if (lane_ready(helper) == false)
        continue;
      if (int rc = allocate_photon_grid(helper)) {
        context->error = helper->error;
        return rc;
      }
    }
  }
  const size_t pixels = size_t(context->pipe.capacity);
  if (noise_threshold > 0.0f) {
    if (context->adaptive_pixels != pixels) {
      int rc = 0;
      if ((rc = device_alloc(context, context->adaptive_sum, pixels)) || (rc = device_alloc(context, context->pixel_state, pixels)))
        return rc;
      context->adaptive_pixels = pixels;
    }
    HIP_OK(context, hipMemsetAsync(context->adaptive_sum, 0, pixels * sizeof(float4), context->stream));
    HIP_OK(context, hipMemsetAsync(context->pixel_state, 0, pixels * sizeof(uint32_t), context->stream));
  }
  context->noise_threshold = noise_threshold;
  context->pipe.adaptive_sum = (noise_threshold > 0.0f) ? context->adaptive_sum : nullptr;
  context->pipe.pixel_state = (noise_threshold > 0.0f) ? context->pixel_state : nullptr;
  context->active_lanes = lanes_wanted;
  context->plans[0] = context->plans[1] = {};  // launch plans belong to a run (integrator, options, scene)
  for (etx_hip_context* helper : context->helpers) {
    helper->plans[0] = helper->plans[1] = {};
    helper->pixel_first = pixel_first, helper->pixel_stride = pixel_stride;
    helper->noise_threshold = noise_threshold;
    helper->pipe.adaptive_sum = context->pipe.adaptive_sum;
    helper->pipe.pixel_state = context->pipe.pixel_state;
    helper->integrator = integrator;
    helper->vcm_options = context->vcm_options;
    helper->pt_options = context->pt_options;
    helper->bdpt_options = context->bdpt_options;
    helper->active_bluenoise = context->active_bluenoise;
  }
  const size_t n = size_t(context->pipe.capacity);
  HIP_OK(context, hipMemsetAsync(context->pipe.camera_sum, 0, n * kFilmLayers * sizeof(float4), context->stream));
  HIP_OK(context, hipMemsetAsync(context->pt_iteration_image, 0, n * 2u * sizeof(float4), context->stream));
  for (etx_hip_context* helper : context->helpers) {
    if (lane_ready(helper))
      HIP_OK(context, hipMemsetAsync(helper->pt_iteration_image, 0, n * 2u * sizeof(float4), context->stream));
  }
  HIP_OK(context, hipStreamSynchronize(context->stream));  // the lanes run on their own streams
  context->armed = true;
  return ETX_HIP_OK;
}

namespace {
// Hands the next iteration to a free lane. `wait`: block while every lane is busy; otherwise return 0 at once.
int submit_iteration(etx_hip_context* context, bool wait) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->armed == false) {
    context->error = "etx_hip_render_iteration: call etx_hip_begin first";
    return ETX_HIP_ERROR_STATE;
  }
  // Adaptive sampling (path tracing with Scene::noise_threshold > 0): the convergence mask an iteration reads is the one the estimate
  // after the previous iteration left (Film::estimate_noise_levels runs serially between iterations, path_tracing.cxx:91-99), so such a
  // render uses ONE lane: iterations neither overlap nor race on the shared mask, and a render is the same every time it runs.
  const bool serial = (context->integrator == ETX_HIP_INTEGRATOR_PT) && (context->noise_threshold > 0.0f);
  const uint32_t lane_count = serial ? 1u : context->active_lanes;
  etx_hip_context* lane = nullptr;
  {
    std::unique_lock<std::mutex> lock(context->shared_mutex);
    if (wait)
      context->idle_cv.wait(lock, [&] { return (context->jobs_in_flight < lane_count) || (context->sticky_error != 0); });
    if (context->sticky_error) {
      context->error = context->sticky_error_text;
      return context->sticky_error;
    }
    if (context->jobs_in_flight >= lane_count)
      return 0;  // every lane is busy (try variant)
    lane = context->lane_busy ? nullptr : context;
    for (size_t i = 0; (serial == false) && (lane == nullptr) && (i + 1u < lane_count); ++i)
      lane = context->helpers[i]->lane_busy ? nullptr : context->helpers[i];
    if (lane == nullptr) {
      context->error = "internal: no free lane";
      return ETX_HIP_ERROR_STATE;
    }
    lane->lane_busy = true;
    if (context->jobs_in_flight == 0u)
      context->busy_since = std::chrono::steady_clock::now();
    context->jobs_in_flight += 1;
    context->totals.current_iteration = context->next_iteration;
  }
  {
    std::lock_guard<std::mutex> lock(lane->lane_mutex);
    lane->jobs.push_back(context->next_iteration);
  }
  lane->lane_cv.notify_one();
  context->next_iteration += context->iteration_stride;
  return 1;
}
}  // namespace

int etx_hip_render_iteration(etx_hip_context* context) {
  const int rc = submit_iteration(context, true);  // blocks only while every lane is busy
  return (rc > 0) ? ETX_HIP_OK : ((rc == 0) ? ETX_HIP_ERROR_STATE : rc);
}

int etx_hip_try_render_iteration(etx_hip_context* context) {
  return submit_iteration(context, false);
}

int etx_hip_poll(etx_hip_context* context) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(context->shared_mutex);
  if (context->sticky_error) {
    context->error = context->sticky_error_text;
    return context->sticky_error;
  }
  return (context->jobs_in_flight == 0u) ? 1 : 0;
}

int etx_hip_sync(etx_hip_context* context) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  int rc = wait_idle(context);
  if (rc)
    return rc;
  HIP_OK(context, hipSetDevice(context->device));
  HIP_OK(context, hipStreamSynchronize(context->stream));
  return ETX_HIP_OK;
}

int etx_hip_read_film(etx_hip_context* context, int layer, float* dst_rgba, size_t dst_bytes) {
  if ((context == nullptr) || (dst_rgba == nullptr))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->scene_ready == false) {
    context->error = "etx_hip_read_film: no scene uploaded";
    return ETX_HIP_ERROR_STATE;
  }
  const size_t n = context->pipe.capacity;
  if (dst_bytes != n * sizeof(float4)) {
    context->error = "etx_hip_read_film: dst_bytes must be width*height*16";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if ((layer < ETX_HIP_LAYER_CAMERA) || (layer > ETX_HIP_LAYER_ALBEDO)) {
    context->error = "etx_hip_read_film: unknown layer";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  HIP_OK(context, hipSetDevice(context->device));
  if (context->comm != nullptr) {
    // a context with a communicator: the whole-job film of the newest reduce, once there is one (a reduce that is still in flight is waited for).
    // Normalised pixel by pixel by the reduced sample count (camera layer's w): ranks need not have finished the same number of iterations.
    if (context->reduce.pending) {
      const int rc = etx_hip_reduce_film_end(context, 1);
      if (rc < 0)
        return rc;
    }
    if (context->reduce.valid && (context->reduce.pixels == n)) {
      const EtxReduceState& r = context->reduce;
      const float4* source = (layer == ETX_HIP_LAYER_NORMAL) ? r.reduced + 2u * n : ((layer == ETX_HIP_LAYER_ALBEDO) ? r.reduced + 3u * n : r.reduced);
      const int mode = (layer == ETX_HIP_LAYER_NORMAL) ? 3 : ((layer == ETX_HIP_LAYER_ALBEDO) ? 0 : layer);
      launch_film_resolve(r.stream, source, r.reduced + n, context->resolve_buffer, uint32_t(n), 0.0f, mode, r.reduced);
      TRANSFER_OK(context, context->transfer.to_host(dst_rgba, context->resolve_buffer, dst_bytes, r.stream, context->error));
      return ETX_HIP_OK;
    }
  }
  int sync_rc = wait_idle(context);
  if (sync_rc)
    return sync_rc;
  uint64_t iterations = context->local_iterations;
  float scale = iterations ? float(1.0 / double(iterations)) : 0.0f;
  // adaptive sampling: pixels hold different sample counts (Film::accumulate_camera_image keeps a running mean per pixel)
  const float4* counts = (context->pipe.pixel_state != nullptr) ? context->pipe.camera_sum : nullptr;
  if (layer == ETX_HIP_LAYER_NORMAL)
    launch_film_resolve(context->stream, context->pipe.normal_sum, context->pipe.light_sum, context->resolve_buffer, uint32_t(n), scale, 3, counts);
  else if (layer == ETX_HIP_LAYER_ALBEDO)
    launch_film_resolve(context->stream, context->pipe.albedo_sum, context->pipe.light_sum, context->resolve_buffer, uint32_t(n), scale, 0, counts);
  else
    launch_film_resolve(context->stream, context->pipe.camera_sum, context->pipe.light_sum, context->resolve_buffer, uint32_t(n), scale, layer, counts);
  TRANSFER_OK(context, context->transfer.to_host(dst_rgba, context->resolve_buffer, dst_bytes, context->stream, context->error));
  return ETX_HIP_OK;
}

// Asynchronous read-back: the resolve kernel and the device-to-host copy run on a stream of their own and
// never wait for the lanes - a GUI host polls from Integrator::update() and keeps its frame rate. The film sums only ever
// hold COMPLETED iterations (every lane commits its iteration image at the end of the iteration), so a snapshot taken
// while other iterations are in flight is a valid progressive image, normalised by the iterations completed so far.
int etx_hip_read_film_begin(etx_hip_context* context, int layer) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->scene_ready == false) {
    context->error = "etx_hip_read_film_begin: no scene uploaded";
    return ETX_HIP_ERROR_STATE;
  }
  if ((layer < ETX_HIP_LAYER_CAMERA) || (layer > ETX_HIP_LAYER_ALBEDO)) {
    context->error = "etx_hip_read_film_begin: unknown layer";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if (context->read_pending) {
    context->error = "etx_hip_read_film_begin: the previous read-back has not been collected (etx_hip_read_film_end)";
    return ETX_HIP_ERROR_STATE;
  }
  HIP_OK(context, hipSetDevice(context->device));
  const size_t n = context->pipe.capacity;
  if (context->read_stream == nullptr) {
    HIP_OK(context, hipStreamCreateWithFlags(&context->read_stream, hipStreamNonBlocking));
    HIP_OK(context, hipEventCreateWithFlags(&context->read_event, hipEventDisableTiming));
  }
  if (context->read_pixels != n) {
    if (context->read_resolve)
      (void)hipFree(context->read_resolve);
    if (context->read_staging)
      (void)hipHostFree(context->read_staging);
    context->read_resolve = nullptr, context->read_staging = nullptr, context->read_pixels = 0;
    HIP_OK(context, hipMalloc(reinterpret_cast<void**>(&context->read_resolve), n * sizeof(float4)));
    HIP_OK(context, hipHostMalloc(reinterpret_cast<void**>(&context->read_staging), n * sizeof(float4), hipHostMallocDefault));
    context->read_pixels = n;
  }
  const int mode = (layer == ETX_HIP_LAYER_NORMAL) ? 3 : ((layer == ETX_HIP_LAYER_ALBEDO) ? 0 : layer);
  if ((context->comm != nullptr) && context->reduce.valid && (context->reduce.pixels == n)) {
    // the reduced copy (whole job): behind the newest reduce on the communication stream - also one that is still in flight, whose result this
    // read-back then returns; the next reduce waits for this read (etx_hip_internal_reduce_prepare)
    const EtxReduceState& r = context->reduce;
    HIP_OK(context, hipStreamWaitEvent(context->read_stream, r.done, 0));
    const float4* reduced_source = (layer == ETX_HIP_LAYER_NORMAL) ? r.reduced + 2u * n : ((layer == ETX_HIP_LAYER_ALBEDO) ? r.reduced + 3u * n : r.reduced);
    launch_film_resolve(context->read_stream, reduced_source, r.reduced + n, context->read_resolve, uint32_t(n), 0.0f, mode, r.reduced);
  } else {
    uint64_t iterations = 0;
    {
      std::lock_guard<std::mutex> lock(context->shared_mutex);
      iterations = uint64_t(context->local_iterations);
    }
    const float scale = iterations ? float(1.0 / double(iterations)) : 0.0f;
    const float4* source = (layer == ETX_HIP_LAYER_NORMAL) ? context->pipe.normal_sum : ((layer == ETX_HIP_LAYER_ALBEDO) ? context->pipe.albedo_sum : context->pipe.camera_sum);
    launch_film_resolve(context->read_stream, source, context->pipe.light_sum, context->read_resolve, uint32_t(n), scale, mode, context->pipe.camera_sum);
  }
  HIP_OK(context, hipMemcpyAsync(context->read_staging, context->read_resolve, n * sizeof(float4), hipMemcpyDeviceToHost, context->read_stream));
  HIP_OK(context, hipEventRecord(context->read_event, context->read_stream));
  context->read_pending = true;
  return ETX_HIP_OK;
}

int etx_hip_read_film_end(etx_hip_context* context, float* dst_rgba, size_t dst_bytes, int wait) {
  if ((context == nullptr) || (dst_rgba == nullptr))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->read_pending == false) {
    context->error = "etx_hip_read_film_end without etx_hip_read_film_begin";
    return ETX_HIP_ERROR_STATE;
  }
  if (dst_bytes != context->read_pixels * sizeof(float4)) {
    context->error = "etx_hip_read_film_end: dst_bytes must be width*height*16";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  HIP_OK(context, hipSetDevice(context->device));
  if (wait) {
    HIP_OK(context, hipEventSynchronize(context->read_event));
  } else {
    const hipError_t q = hipEventQuery(context->read_event);
    if (q == hipErrorNotReady)
      return 0;
    HIP_OK(context, q);
  }
  memcpy(dst_rgba, context->read_staging, dst_bytes);
  context->read_pending = false;
  return 1;
}

namespace {
// Film state of a render in progress (etx_hip_checkpoint_*): this header, the four float4 sum layers (camera with the per-pixel
// sample count in w, light, normal, albedo), and for an adaptive run the even-sample sums and the pixel states.
struct CheckpointHeader {
  uint32_t magic, version, integrator, width, height, first_iteration, iteration_stride, next_iteration;
  uint32_t local_iterations, adaptive, options_hash, last_active_pixels, scene_hash;
  uint32_t pixel_first, pixel_stride_minus_1;  // pixel sharding (etx_hip_begin_ex); zero = every pixel, which is what version-2 checkpoints written before it hold
  uint32_t reserved;
};
static_assert(sizeof(CheckpointHeader) == 64, "checkpoint header layout");
constexpr uint32_t kCheckpointMagic = 0x43585445u;  // "ETXC"
constexpr uint32_t kCheckpointVersion = 2u;  // 2: scene_hash (a formerly reserved word) names the scene, hashed field by field (host_scene.cpp content_hash)

// the named fields only: the padding of the by-value option structs is whatever the caller's stack held
uint32_t options_hash(const etx_hip_context* c) {
  uint32_t h = 2166136261u;  // FNV-1a
  auto mix = [&h](uint32_t v) {
    for (uint32_t i = 0; i < 4u; ++i)
      h = (h ^ ((v >> (8u * i)) & 0xffu)) * 16777619u;
  };
  if (c->integrator == ETX_HIP_INTEGRATOR_PT) {
    const auto& o = c->pt_options;
    mix(o.path_per_iteration), mix(o.nee != 0), mix(o.direct != 0), mix(o.mis != 0), mix(o.blue_noise != 0);
  } else if (c->integrator == ETX_HIP_INTEGRATOR_BDPT) {
    const auto& o = c->bdpt_options;
    mix(o.mode), mix(o.direct_hit != 0), mix(o.connect_to_camera != 0), mix(o.connect_to_light != 0), mix(o.connect_vertices != 0), mix(o.mis != 0), mix(o.blue_noise != 0);
    if (o.reference_seeding)
      mix(0x73656564u);  // only when set: checkpoints written before the field existed keep their hash
  } else {
    const auto& o = c->vcm_options;
    uint32_t radius_bits = 0;
    memcpy(&radius_bits, &o.initial_radius, sizeof(radius_bits));
    mix(o.options), mix(o.radius_decay), mix(o.kernel), mix(radius_bits), mix(o.blue_noise != 0);
    if (o.reference_seeding)
      mix(0x73656564u);
  }
  return h;
}

// What a film is a film OF, as far as the uploaded scene tells: the scalars the integrators read, the camera, the sizes of the tables
// (a checkpoint does not hold the scene; this keeps it from being continued on another one by mistake)
uint32_t scene_hash(const etx_hip_context* c) {
  const DScene& sc = c->scene.host_copy;
  uint32_t h = 2166136261u;
  auto mix_bytes = [&h](const void* data, size_t size) {
    const uint8_t* b = static_cast<const uint8_t*>(data);
    for (size_t i = 0; i < size; ++i)
      h = (h ^ b[i]) * 16777619u;
  };
  const uint32_t words[] = {sc.vertex_count, sc.triangle_count, sc.material_count, sc.emitter_count, sc.medium_count, sc.image_count, sc.spectrum_count, sc.min_path_length, sc.max_path_length,
    sc.samples, sc.random_path_termination, sc.flags, sc.spectral};
  mix_bytes(words, sizeof(words));
  mix_bytes(&sc.radiance_clamp, sizeof(float));
  mix_bytes(&sc.bounds_center, sizeof(sc.bounds_center));
  mix_bytes(&sc.bounds_radius, sizeof(float));
  mix_bytes(&sc.camera, sizeof(sc.camera));
  mix_bytes(&c->scene.content_hash, sizeof(uint32_t));
  return h;
}

size_t checkpoint_bytes(const etx_hip_context* c) {
  const size_t n = c->pipe.capacity;
  return sizeof(CheckpointHeader) + n * kFilmLayers * sizeof(float4) + ((c->pipe.pixel_state != nullptr) ? n * (sizeof(float4) + sizeof(uint32_t)) : 0u);
}
}  // namespace

size_t etx_hip_checkpoint_bytes(const etx_hip_context* context) {
  return ((context == nullptr) || (context->armed == false)) ? 0u : checkpoint_bytes(context);
}

int etx_hip_checkpoint_save(etx_hip_context* context, void* dst, size_t dst_bytes) {
  if ((context == nullptr) || (dst == nullptr))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->armed == false) {
    context->error = "etx_hip_checkpoint_save: call etx_hip_begin first (a checkpoint holds this rank's own sums; film reduces do not touch them)";
    return ETX_HIP_ERROR_STATE;
  }
  if (dst_bytes != checkpoint_bytes(context)) {
    context->error = "etx_hip_checkpoint_save: dst_bytes must be etx_hip_checkpoint_bytes()";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if (int rc = wait_idle(context))
    return rc;
  HIP_OK(context, hipSetDevice(context->device));
  const size_t n = context->pipe.capacity;
  CheckpointHeader header = {};
  header.magic = kCheckpointMagic, header.version = kCheckpointVersion, header.integrator = uint32_t(context->integrator);
  header.width = context->scene.film_w, header.height = context->scene.film_h;
  header.first_iteration = context->first_iteration, header.iteration_stride = context->iteration_stride;
  header.pixel_first = context->pixel_first, header.pixel_stride_minus_1 = context->pixel_stride - 1u;
  header.next_iteration = context->next_iteration, header.local_iterations = context->local_iterations;
  header.adaptive = (context->pipe.pixel_state != nullptr) ? 1u : 0u;
  header.options_hash = options_hash(context);
  header.scene_hash = scene_hash(context);
  {
    std::lock_guard<std::mutex> lock(context->shared_mutex);
    header.last_active_pixels = context->totals.last_active_pixels;  // the path tracer's host stops on "the last iteration sampled no pixel"
  }
  uint8_t* out = static_cast<uint8_t*>(dst);
  memcpy(out, &header, sizeof(header));
  out += sizeof(header);
  TRANSFER_OK(context, context->transfer.to_host(out, context->pipe.camera_sum, n * kFilmLayers * sizeof(float4), context->stream, context->error));
  out += n * kFilmLayers * sizeof(float4);
  if (header.adaptive) {
    TRANSFER_OK(context, context->transfer.to_host(out, context->pipe.adaptive_sum, n * sizeof(float4), context->stream, context->error));
    TRANSFER_OK(context, context->transfer.to_host(out + n * sizeof(float4), context->pipe.pixel_state, n * sizeof(uint32_t), context->stream, context->error));
  }
  return ETX_HIP_OK;
}

int etx_hip_checkpoint_load(etx_hip_context* context, const void* src, size_t src_bytes) {
  if ((context == nullptr) || (src == nullptr))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->armed == false) {
    context->error = "etx_hip_checkpoint_load: call etx_hip_begin (same integrator, options and iteration sharding as the saved run) first";
    return ETX_HIP_ERROR_STATE;
  }
  if (int rc = wait_idle(context))
    return rc;
  CheckpointHeader header = {};
  if (src_bytes >= sizeof(header))
    memcpy(&header, src, sizeof(header));
  if ((src_bytes < sizeof(header)) || (header.magic != kCheckpointMagic) || (header.version != kCheckpointVersion)) {
    context->error = "etx_hip_checkpoint_load: not a checkpoint of this library version (expected version " + std::to_string(kCheckpointVersion) + ")";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  const bool adaptive = context->pipe.pixel_state != nullptr;
  if ((header.integrator != uint32_t(context->integrator)) || (header.width != context->scene.film_w) || (header.height != context->scene.film_h) ||
      (header.first_iteration != context->first_iteration) || (header.iteration_stride != context->iteration_stride) || (header.pixel_first != context->pixel_first) ||
      (header.pixel_stride_minus_1 != context->pixel_stride - 1u) || ((header.adaptive != 0u) != adaptive) ||
      (header.options_hash != options_hash(context)) || (header.scene_hash != scene_hash(context)) || (src_bytes != checkpoint_bytes(context))) {
    context->error = "etx_hip_checkpoint_load: the checkpoint was saved by a different run (integrator, options, scene scalars / camera / table sizes, film size, adaptive sampling or iteration / pixel sharding differ)";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  if (uint64_t(header.next_iteration) != uint64_t(header.first_iteration) + uint64_t(header.local_iterations) * uint64_t(header.iteration_stride)) {
    context->error = "etx_hip_checkpoint_load: inconsistent iteration counters in the checkpoint header";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  HIP_OK(context, hipSetDevice(context->device));
  const size_t n = context->pipe.capacity;
  const uint8_t* in = static_cast<const uint8_t*>(src) + sizeof(header);
  TRANSFER_OK(context, context->transfer.to_device(context->pipe.camera_sum, in, n * kFilmLayers * sizeof(float4), context->stream, context->error));
  in += n * kFilmLayers * sizeof(float4);
  if (adaptive) {
    TRANSFER_OK(context, context->transfer.to_device(context->pipe.adaptive_sum, in, n * sizeof(float4), context->stream, context->error));
    TRANSFER_OK(context, context->transfer.to_device(context->pipe.pixel_state, in + n * sizeof(float4), n * sizeof(uint32_t), context->stream, context->error));
  }
  context->next_iteration = header.next_iteration;
  context->local_iterations = header.local_iterations;
  context->reduce.valid = false;  // a reduced copy of the film before the load no longer describes this context's run
  {
    std::lock_guard<std::mutex> lock(context->shared_mutex);
    context->totals = {};
    context->totals.completed_iterations = header.local_iterations;
    context->totals.current_iteration = header.next_iteration;
    context->totals.last_active_pixels = header.last_active_pixels;
  }
  return ETX_HIP_OK;
}

size_t etx_hip_device_bytes(const etx_hip_context* context) {
  if (context == nullptr)
    return 0;
  size_t bytes = context->allocated_bytes;
  for (const etx_hip_context* helper : context->helpers)
    bytes += helper->allocated_bytes;
  return bytes;
}

int etx_hip_set_debug_flags(etx_hip_context* context, uint32_t flags) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (int rc = wait_idle(context))
    return rc;
  context->debug_flags = context->pipe.debug_flags = flags;
  for (etx_hip_context* helper : context->helpers)
    helper->debug_flags = helper->pipe.debug_flags = flags;
  return ETX_HIP_OK;
}

int etx_hip_set_timers(etx_hip_context* context, uint32_t mask) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  (void)wait_idle(context);
  context->timer_mask = mask;
  for (etx_hip_context* helper : context->helpers)
    helper->timer_mask = mask;
  return ETX_HIP_OK;
}

int etx_hip_stats(etx_hip_context* context, etx_hip_stats_t* out_stats, size_t stats_size) {
  // a client built against an earlier header asks for the prefix it knows - exactly one of the layouts that ever shipped (fields are only
  // appended): ABI 1 ended before pool_grows, ABI 2 and later hold the whole struct. Any other size is a binding that has drifted.
  const bool known_layout = (stats_size == offsetof(etx_hip_stats_t, pool_grows)) || (stats_size == sizeof(etx_hip_stats_t));
  if ((context == nullptr) || (out_stats == nullptr) || (known_layout == false))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(context->shared_mutex);
  memcpy(out_stats, &context->totals, stats_size);
  return ETX_HIP_OK;
}

int etx_hip_trace_rays_timed(etx_hip_context* context, const float* rays_8f, uint64_t count, uint32_t repeat, double* out_avg_ms, float* hits_4f) {
  if ((context == nullptr) || ((count > 0) && (rays_8f == nullptr)))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->scene_ready == false) {
    context->error = "etx_hip_trace_rays: no scene uploaded";
    return ETX_HIP_ERROR_STATE;
  }
  if (out_avg_ms)
    *out_avg_ms = 0.0;
  if (count == 0)
    return ETX_HIP_OK;
  if (count > 0xffffffffull) {
    context->error = "ray count exceeds 2^32";
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  }
  HIP_OK(context, hipSetDevice(context->device));
  std::vector<float4> o(count), d(count);
  for (uint64_t i = 0; i < count; ++i) {
    o[i] = make_float4(rays_8f[8 * i + 0], rays_8f[8 * i + 1], rays_8f[8 * i + 2], rays_8f[8 * i + 3]);
    d[i] = make_float4(rays_8f[8 * i + 4], rays_8f[8 * i + 5], rays_8f[8 * i + 6], rays_8f[8 * i + 7]);
  }
  float4 *d_o = nullptr, *d_d = nullptr, *d_h = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = ETX_HIP_OK;
  auto fail = [&](const std::string& what, hipError_t e) {
    context->error = what + ": " + hipGetErrorString(e);
    rc = ETX_HIP_ERROR_HIP;
  };
  do {
    hipError_t e = hipSuccess;
    if (((e = hipMalloc(reinterpret_cast<void**>(&d_o), count * sizeof(float4))) != hipSuccess) || ((e = hipMalloc(reinterpret_cast<void**>(&d_d), count * sizeof(float4))) != hipSuccess) ||
        ((e = hipMalloc(reinterpret_cast<void**>(&d_h), count * sizeof(float4))) != hipSuccess)) {
      fail("hipMalloc of the ray queues", e);
      break;
    }
    if ((rc = context->transfer.to_device(d_o, o.data(), count * sizeof(float4), context->stream, context->error)) ||
        (rc = context->transfer.to_device(d_d, d.data(), count * sizeof(float4), context->stream, context->error)))
      break;
    const bool flat = context->scene.host_copy.bvh_flat != 0u;
    launch_trace_rays(context->stream, context->pipe.scene, d_o, d_d, d_h, uint32_t(count), flat, context->debug_flags);
    if (repeat > 0u) {  // the first launch was the untimed one (code object load, caches)
      if (((e = hipEventCreate(&e0)) != hipSuccess) || ((e = hipEventCreate(&e1)) != hipSuccess) || ((e = hipEventRecord(e0, context->stream)) != hipSuccess)) {
        fail("hipEvent", e);
        break;
      }
      for (uint32_t r = 0; r < repeat; ++r)
        launch_trace_rays(context->stream, context->pipe.scene, d_o, d_d, d_h, uint32_t(count), flat, context->debug_flags);
      float ms = 0.0f;
      if (((e = hipEventRecord(e1, context->stream)) != hipSuccess) || ((e = hipEventSynchronize(e1)) != hipSuccess) || ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess)) {
        fail("trace kernel failed", e);
        break;
      }
      if (out_avg_ms)
        *out_avg_ms = double(ms) / double(repeat);
    }
    if (hits_4f != nullptr) {
      if ((rc = context->transfer.to_host(hits_4f, d_h, count * sizeof(float4), context->stream, context->error))) {
        context->error = "trace kernel failed: " + context->error;
        break;
      }
    } else if ((e = hipStreamSynchronize(context->stream)) != hipSuccess) {
      fail("trace kernel failed", e);
    }
  } while (false);
  if (e0)
    (void)hipEventDestroy(e0);
  if (e1)
    (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(context->stream);
  (void)hipFree(d_o);
  (void)hipFree(d_d);
  (void)hipFree(d_h);
  return rc;
}

int etx_hip_trace_rays(etx_hip_context* context, const float* rays_8f, uint64_t count, float* hits_4f) {
  if ((count > 0) && (hits_4f == nullptr))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  return etx_hip_trace_rays_timed(context, rays_8f, count, 0u, nullptr, hits_4f);
}

int etx_hip_trace_rays_device(etx_hip_context* context, const void* d_rays_o_tmin, const void* d_rays_d_tmax, uint64_t count, void* d_hits, uint32_t repeat, double* out_avg_ms) {
  if ((context == nullptr) || (d_rays_o_tmin == nullptr) || (d_rays_d_tmax == nullptr) || (d_hits == nullptr) || (count == 0) || (count > 0xffffffffull) || (repeat == 0))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (context->scene_ready == false) {
    context->error = "etx_hip_trace_rays_device: no scene uploaded";
    return ETX_HIP_ERROR_STATE;
  }
  HIP_OK(context, hipSetDevice(context->device));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_OK(context, hipEventCreate(&e0));
  HIP_OK(context, hipEventCreate(&e1));
  // one untimed launch (code object load, caches)
  launch_trace_rays(context->stream, context->pipe.scene, reinterpret_cast<const float4*>(d_rays_o_tmin), reinterpret_cast<const float4*>(d_rays_d_tmax),
    reinterpret_cast<float4*>(d_hits), uint32_t(count), context->scene.host_copy.bvh_flat != 0u, context->debug_flags);
  HIP_OK(context, hipEventRecord(e0, context->stream));
  for (uint32_t r = 0; r < repeat; ++r)
    launch_trace_rays(context->stream, context->pipe.scene, reinterpret_cast<const float4*>(d_rays_o_tmin), reinterpret_cast<const float4*>(d_rays_d_tmax),
      reinterpret_cast<float4*>(d_hits), uint32_t(count), context->scene.host_copy.bvh_flat != 0u, context->debug_flags);
  HIP_OK(context, hipEventRecord(e1, context->stream));
  HIP_OK(context, hipEventSynchronize(e1));
  float ms = 0.0f;
  HIP_OK(context, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (out_avg_ms)
    *out_avg_ms = double(ms) / double(repeat);
  return ETX_HIP_OK;
}

int etx_hip_selftest_stack(etx_hip_context* context, uint32_t depth, uint32_t* out_errors) {
  if ((context == nullptr) || (out_errors == nullptr) || (depth == 0u) || (depth > kMaxStackDepth))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  HIP_OK(context, hipSetDevice(context->device));
  const uint32_t blocks = 1024u, lanes = blocks * kBlockSize;
  int32_t* spill = nullptr;
  uint32_t* errors = nullptr;
  HIP_OK(context, hipMalloc(&spill, size_t(lanes) * (std::max(depth, 64u) - kShortStackDepth) * sizeof(int32_t)));  // rows of the short stack cover the other's
  if (hipMalloc(&errors, sizeof(uint32_t)) != hipSuccess) {
    (void)hipFree(spill);
    context->error = "hipMalloc failed";
    return ETX_HIP_ERROR_HIP;
  }
  (void)hipMemsetAsync(errors, 0, sizeof(uint32_t), context->stream);
  launch_stack_selftest(context->stream, spill, lanes, blocks, depth, errors);
  const int copied = context->transfer.to_host(out_errors, errors, sizeof(uint32_t), context->stream, context->error);
  const hipError_t synced = hipStreamSynchronize(context->stream);
  (void)hipFree(spill);
  (void)hipFree(errors);
  if (copied)
    return copied;
  HIP_OK(context, synced);
  return ETX_HIP_OK;
}

int etx_hip_kat(etx_hip_context* context, int which, const float* in, uint64_t count, float* out) {
  static const uint32_t in_width[] = {2, 6, 3, 5, 4, 2, 3};
  static const uint32_t out_width[] = {4, 3, 6, 3, 1, 2, 6};
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  // 6 + 16 * set_index: blue-noise lookups in the uploaded table of that sample-count class
  const uint2* bluenoise = nullptr;
  if ((which >= 6) && ((which & 15) == 6) && ((which >> 4) < int(kBlueNoiseSets))) {
    bluenoise = reinterpret_cast<const uint2*>(context->bluenoise[which >> 4]);
    which = 6;
    if (bluenoise == nullptr) {
      context->error = "etx_hip_kat: that blue-noise set has not been uploaded";
      return ETX_HIP_ERROR_STATE;
    }
  }
  if ((which < 0) || (which > 6) || ((which == 6) && (bluenoise == nullptr)) || (in == nullptr) || (out == nullptr) || (count > (1u << 24)))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  if (count == 0)
    return ETX_HIP_OK;
  HIP_OK(context, hipSetDevice(context->device));
  float *d_in = nullptr, *d_out = nullptr;
  HIP_OK(context, hipMalloc(reinterpret_cast<void**>(&d_in), count * in_width[which] * sizeof(float)));
  HIP_OK(context, hipMalloc(reinterpret_cast<void**>(&d_out), count * out_width[which] * sizeof(float)));
  int rc = ETX_HIP_OK;
  rc = context->transfer.to_device(d_in, in, count * in_width[which] * sizeof(float), context->stream, context->error);
  if (rc == ETX_HIP_OK) {
    launch_kat(context->stream, which, d_in, uint32_t(count), d_out, bluenoise);
    rc = context->transfer.to_host(out, d_out, count * out_width[which] * sizeof(float), context->stream, context->error);
  }
  if (rc)
    context->error = "etx_hip_kat failed: " + context->error;
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc;
}

int etx_hip_host_check_bvh(const etx_abi_scene* scene, uint32_t out_info[4]) {
  return etx_hip_host_check_bvh_builder(scene, ETX_HIP_BVH_HOST_SAH, out_info);
}

int etx_hip_host_check_bvh_builder(const etx_abi_scene* scene, int builder, uint32_t out_info[4]) {
  if ((scene == nullptr) || (out_info == nullptr) || ((builder != ETX_HIP_BVH_HOST_SAH) && (builder != ETX_HIP_BVH_DEVICE_LBVH)))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  etxh::HostBvh bvh;
  if (builder == ETX_HIP_BVH_DEVICE_LBVH)
    etxh::build_lbvh_host(scene, bvh);  // the device build, emulated element by element
  else
    etxh::build_bvh(scene, bvh);
  const auto* vertices = reinterpret_cast<const etx_abi_vertex*>(scene->vertices.a);
  const auto* triangles = reinterpret_cast<const etx_abi_triangle*>(scene->triangles.a);
  const uint32_t n = uint32_t(scene->triangles.count);
  std::vector<uint32_t> seen(n, 0u);
  bool ok = bvh.tris.size() == n;
  auto bits = [](float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
  };
  // recursive descent with an explicit stack: (child reference, enclosing box)
  struct Item {
    int32_t ref;
    f3 lo, hi;
  };
  std::vector<Item> stack;
  if ((n > 0) && (bvh.nodes.empty() == false))  // the builder's intermediate BVH2 (the binned-SAH build keeps it; a one-leaf scene and the linear build have none)
    stack.push_back({bvh.root, mk3(-kMaxFloat), mk3(kMaxFloat)});
  while (ok && (stack.empty() == false)) {
    Item it = stack.back();
    stack.pop_back();
    if (it.ref >= 0) {
      if (size_t(it.ref) >= bvh.nodes.size()) {
        ok = false;
        break;
      }
      const BvhNode& nd = bvh.nodes[it.ref];
      stack.push_back({nd.child0, {nd.lo0_hi0x.x, nd.lo0_hi0x.y, nd.lo0_hi0x.z}, {nd.lo0_hi0x.w, nd.hi0yz_lo1xy.x, nd.hi0yz_lo1xy.y}});
      stack.push_back({nd.child1, {nd.hi0yz_lo1xy.z, nd.hi0yz_lo1xy.w, nd.lo1z_hi1.x}, {nd.lo1z_hi1.y, nd.lo1z_hi1.z, nd.lo1z_hi1.w}});
    } else {
      uint32_t leaf = uint32_t(~it.ref), first = leaf >> 3, count = (leaf & 7u) + 1u;
      for (uint32_t i = first; ok && (i < first + count); ++i) {
        if (i >= n) {
          ok = false;
          break;
        }
        uint32_t ti = bits(bvh.tris[i].v0_index.w);
        if (ti >= n) {
          ok = false;
          break;
        }
        seen[ti]++;
        for (int k = 0; k < 3; ++k) {
          const etx_abi_float3& p = vertices[triangles[ti].i[k]].pos;
          ok = ok && (p.x >= it.lo.x) && (p.y >= it.lo.y) && (p.z >= it.lo.z) && (p.x <= it.hi.x) && (p.y <= it.hi.y) && (p.z <= it.hi.z);
        }
      }
    }
  }
  for (uint32_t i = 0; ok && (bvh.nodes.empty() == false) && (i < n); ++i)
    ok = seen[i] == 1u;
  // the BVH4 the device traverses: the same invariants (every triangle in exactly one leaf, children inside their boxes),
  // children numbered breadth first (child index > parent index)
  std::vector<uint32_t> seen4(n, 0u);
  struct Item4 {
    int32_t ref;
    f3 lo, hi;
  };
  std::vector<Item4> stack4;
  if (n > 0)
    stack4.push_back({bvh.root4, mk3(-kMaxFloat), mk3(kMaxFloat)});
  while (ok && (stack4.empty() == false)) {
    const Item4 it = stack4.back();
    stack4.pop_back();
    if (it.ref >= 0) {
      if (size_t(it.ref) >= bvh.nodes4.size()) {
        ok = false;
        break;
      }
      const Bvh4Node& nd = bvh.nodes4[it.ref];
      const float* lo[3] = {&nd.lo_x.x, &nd.lo_y.x, &nd.lo_z.x};
      const float* hi[3] = {&nd.hi_x.x, &nd.hi_y.x, &nd.hi_z.x};
      for (int k = 0; k < 4; ++k) {
        if (nd.child[k] == kBvhEmptyChild)
          continue;
        ok = ok && ((nd.child[k] < 0) || (nd.child[k] > it.ref));
        const f3 clo = {lo[0][k], lo[1][k], lo[2][k]}, chi = {hi[0][k], hi[1][k], hi[2][k]};
        ok = ok && (clo.x >= it.lo.x) && (clo.y >= it.lo.y) && (clo.z >= it.lo.z) && (chi.x <= it.hi.x) && (chi.y <= it.hi.y) && (chi.z <= it.hi.z);
        stack4.push_back({nd.child[k], clo, chi});
      }
    } else {
      const uint32_t leaf = uint32_t(~it.ref), first = leaf >> 3, count = (leaf & 7u) + 1u;
      for (uint32_t i = first; ok && (i < first + count); ++i) {
        if (i >= n) {
          ok = false;
          break;
        }
        const uint32_t ti = bits(bvh.tris[i].v0_index.w);
        if (ti >= n) {
          ok = false;
          break;
        }
        seen4[ti]++;
        for (int k = 0; k < 3; ++k) {
          const etx_abi_float3& p = vertices[triangles[ti].i[k]].pos;
          ok = ok && (p.x >= it.lo.x) && (p.y >= it.lo.y) && (p.z >= it.lo.z) && (p.x <= it.hi.x) && (p.y <= it.hi.y) && (p.z <= it.hi.z);
        }
      }
    }
  }
  for (uint32_t i = 0; ok && (i < n); ++i)
    ok = seen4[i] == 1u;
  ok = ok && (bvh.stack_need <= etxd::kMaxStackDepth);
  out_info[0] = uint32_t(bvh.nodes4.size());
  out_info[1] = uint32_t(bvh.tris.size());
  out_info[2] = bvh.depth4 | (bvh.stack_need << 16u);
  out_info[3] = uint32_t(bvh.nodes4.size() * sizeof(Bvh4Node) + bvh.tris.size() * sizeof(BvhTri));
  return ok ? ETX_HIP_OK : ETX_HIP_ERROR_INVALID_ARGUMENT;
}

// Host-only: walks the BVH4 exactly as dev_bvh.h bvh_closest does (near child first, three pushes per node at most) for
// `count` rays {ox,oy,oz,tmin,dx,dy,dz,tmax} and reports the work: out[0] node visits, out[1] triangle tests, out[2] rays
// that hit, out[3] deepest stack use. Used by tests (stack bound) and to reason about the traversal kernel's cost.
int etx_hip_host_bvh_stats(const etx_abi_scene* scene, const float* rays_8f, uint64_t count, uint64_t out[4]) {
  return etx_hip_host_bvh_stats_builder(scene, ETX_HIP_BVH_HOST_SAH, rays_8f, count, out, nullptr);
}

int etx_hip_host_bvh_stats_builder(const etx_abi_scene* scene, int builder, const float* rays_8f, uint64_t count, uint64_t out[4], float* hits_2f) {
  if ((scene == nullptr) || (rays_8f == nullptr) || (out == nullptr) || ((builder != ETX_HIP_BVH_HOST_SAH) && (builder != ETX_HIP_BVH_DEVICE_LBVH)))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  etxh::HostBvh bvh;
  if (builder == ETX_HIP_BVH_DEVICE_LBVH)
    etxh::build_lbvh_host(scene, bvh);
  else
    etxh::build_bvh(scene, bvh);
  out[0] = out[1] = out[2] = out[3] = 0;
  if (bvh.tris.empty())
    return ETX_HIP_OK;
  std::vector<int32_t> stack(256);
  for (uint64_t r = 0; r < count; ++r) {
    const float* q = rays_8f + 8 * r;
    const f3 o = {q[0], q[1], q[2]}, d = {q[4], q[5], q[6]};
    const float tmin = q[3];
    float best = q[7];
    bool hit = false;
    float hit_triangle = 0.0f;  // index as float bits
    const f3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    size_t sp = 0;
    int32_t cur = bvh.root4;
    while (cur != kBvhEmptyChild) {
      if (cur >= 0) {
        out[0]++;
        const Bvh4Node& nd = bvh.nodes4[cur];
        const float* lo[3] = {&nd.lo_x.x, &nd.lo_y.x, &nd.lo_z.x};
        const float* hi[3] = {&nd.hi_x.x, &nd.hi_y.x, &nd.hi_z.x};
        float t[4];
        int32_t c[4];
        for (int k = 0; k < 4; ++k) {
          c[k] = nd.child[k];
          t[k] = kMaxFloat;
          if (c[k] == kBvhEmptyChild)
            continue;
          float t_enter = tmin, t_exit = best;
          const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
          for (int a = 0; a < 3; ++a) {
            const float t0 = (lo[a][k] - oo[a]) * ii[a], t1 = (hi[a][k] - oo[a]) * ii[a];
            t_enter = std::max(t_enter, std::min(t0, t1));
            t_exit = std::min(t_exit, std::max(t0, t1));
          }
          if (t_enter <= t_exit * 1.0000004f)
            t[k] = t_enter;
        }
        for (int i = 0; i < 4; ++i)  // ascending by t (the device uses a five-comparator network: same order up to ties)
          for (int j = i + 1; j < 4; ++j)
            if (t[j] < t[i]) {
              std::swap(t[i], t[j]);
              std::swap(c[i], c[j]);
            }
        if (t[0] == kMaxFloat) {
          cur = sp ? stack[--sp] : kBvhEmptyChild;
        } else {
          for (int k = 3; k >= 1; --k)
            if (t[k] < kMaxFloat) {
              if (sp == stack.size())
                stack.resize(stack.size() * 2);
              stack[sp++] = c[k];
            }
          out[3] = std::max<uint64_t>(out[3], sp);
          cur = c[0];
        }
      } else {
        const uint32_t leaf = uint32_t(~cur), first = leaf >> 3, n = (leaf & 7u) + 1u;
        for (uint32_t i = first; i < first + n; ++i) {
          out[1]++;
          const BvhTri& tr = bvh.tris[i];
          const f3 e1 = {tr.e1_flags.x, tr.e1_flags.y, tr.e1_flags.z}, e2 = {tr.e2_mat.x, tr.e2_mat.y, tr.e2_mat.z};
          const f3 p = cross(d, e2);
          const float det = dot(e1, p);
          if (det == 0.0f)
            continue;
          const f3 s = o - f3{tr.v0_index.x, tr.v0_index.y, tr.v0_index.z};
          const float u = dot(s, p) / det;
          const f3 qv = cross(s, e1);
          const float v = dot(d, qv) / det, tt = dot(e2, qv) / det;
          if ((u >= 0.0f) && (v >= 0.0f) && (u + v <= 1.0f) && (tt >= tmin) && (tt <= best)) {
            best = tt;
            hit = true;
            hit_triangle = tr.v0_index.w;
          }
        }
        cur = sp ? stack[--sp] : kBvhEmptyChild;
      }
    }
    out[2] += hit ? 1u : 0u;
    if (hits_2f != nullptr) {
      const uint32_t miss = 0xffffffffu;
      hits_2f[2 * r + 0] = hit ? best : 0.0f;
      if (hit)
        hits_2f[2 * r + 1] = hit_triangle;
      else
        memcpy(hits_2f + 2 * r + 1, &miss, 4);
    }
  }
  return ETX_HIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The context's half of the multi-GPU film reduce (host_reduce.h; the RCCL half is host_comm.cpp). Kept out of the public header.
hipStream_t etx_hip_internal_stream(etx_hip_context* c) {
  return c->stream;
}
void** etx_hip_internal_comm(etx_hip_context* c) {
  return &c->comm;
}
EtxReduceState* etx_hip_internal_reduce(etx_hip_context* c) {
  return &c->reduce;
}
void etx_hip_internal_set_error(etx_hip_context* c, const std::string& e) {
  c->error = e;
}
void etx_hip_internal_rank(etx_hip_context* c, int** rank, int** world) {
  *rank = &c->rank;
  *world = &c->world;
}
uint64_t* etx_hip_internal_global_iterations(etx_hip_context* c) {
  return &c->global_iterations;
}
int etx_hip_internal_device(etx_hip_context* c) {
  return c->device;
}

namespace {
void free_reduce_buffers(EtxReduceState& r) {
  if (r.snapshot)
    (void)hipFree(r.snapshot);
  if (r.reduced)
    (void)hipFree(r.reduced);
  r.snapshot = r.reduced = nullptr;
  r.pixels = 0;
}
}  // namespace

// Stream, events, counters and - once the film size is known - the two film-sized buffers of the reduce. Called by etx_hip_comm_init (so that a
// rank finds out there, not inside a collective, that it cannot join), by etx_hip_begin on a context that has a communicator, and by the
// reduce itself as a last resort.
int etx_hip_internal_reduce_allocate(etx_hip_context* c) {
  EtxReduceState& r = c->reduce;
  HIP_OK(c, hipSetDevice(c->device));
  if (r.stream == nullptr) {
    HIP_OK(c, hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    HIP_OK(c, hipEventCreateWithFlags(&r.snapshot_done, hipEventDisableTiming));
    HIP_OK(c, hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
    HIP_OK(c, hipEventCreate(&r.time_begin));
    HIP_OK(c, hipEventCreate(&r.time_end));
    HIP_OK(c, hipMalloc(reinterpret_cast<void**>(&r.d_counters), 4 * sizeof(unsigned long long)));
    HIP_OK(c, hipHostMalloc(reinterpret_cast<void**>(&r.h_counters), 4 * sizeof(unsigned long long), hipHostMallocDefault));
    memset(r.h_counters, 0, 4 * sizeof(unsigned long long));
  }
  const size_t n = c->scene_ready ? size_t(c->pipe.capacity) : 0u;
  if ((n != 0u) && (r.pixels != n)) {
    HIP_OK(c, hipStreamSynchronize(r.stream));
    free_reduce_buffers(r);
    r.valid = false;
    HIP_OK(c, hipMalloc(reinterpret_cast<void**>(&r.snapshot), n * kFilmLayers * sizeof(float4)));
    HIP_OK(c, hipMalloc(reinterpret_cast<void**>(&r.reduced), n * kFilmLayers * sizeof(float4)));
    r.pixels = n;
    HIP_OK(c, hipMemsetAsync(r.snapshot, 0, n * kFilmLayers * sizeof(float4), r.stream));
    HIP_OK(c, hipMemsetAsync(r.reduced, 0, n * kFilmLayers * sizeof(float4), r.stream));
    HIP_OK(c, hipStreamSynchronize(r.stream));
  }
  return ETX_HIP_OK;
}

void etx_hip_internal_reduce_release(etx_hip_context* c) {
  EtxReduceState& r = c->reduce;
  {
    std::lock_guard<std::mutex> lock(r.mutex);  // CommitSection reads the flag on the lanes' threads: no commit waits for an event destroyed below
    r.snapshot_recorded = false;
  }
  if (r.stream)
    (void)hipStreamSynchronize(r.stream);
  free_reduce_buffers(r);
  if (r.d_counters)
    (void)hipFree(r.d_counters);
  if (r.h_counters)
    (void)hipHostFree(r.h_counters);
  if (r.h_words)
    (void)hipHostFree(r.h_words);
  if (r.d_words)
    (void)hipFree(r.d_words);
  r.h_words = r.d_words = nullptr;
  for (hipEvent_t* e : {&r.snapshot_done, &r.done, &r.time_begin, &r.time_end}) {
    if (*e)
      (void)hipEventDestroy(*e);
    *e = nullptr;
  }
  if (r.stream)
    (void)hipStreamDestroy(r.stream);
  r.stream = nullptr, r.d_counters = nullptr, r.h_counters = nullptr;
  r.pending = 0u;
  r.valid = false;
}

// etx_hip_begin: the reduced copy belongs to the run that produced it. A reduce still in flight is waited for (every rank begins the same
// runs, so the collective completes); layers the new run's integrator does not write must read as zero.
int etx_hip_internal_reduce_reset(etx_hip_context* c) {
  EtxReduceState& r = c->reduce;
  if (r.stream == nullptr)
    return ETX_HIP_OK;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(r.stream));
  r.pending = 0u, r.valid = false;
  r.pending_local_rc = 0, r.pending_local_error.clear();
  {
    std::lock_guard<std::mutex> lock(r.mutex);
    r.snapshot_recorded = false;  // nothing of the new run has to wait for a snapshot of the old one (the stream is idle)
    c->commit_recorded = false;
    for (etx_hip_context* helper : c->helpers)
      helper->commit_recorded = false;
  }
  if (c->comm != nullptr) {
    if (int rc = etx_hip_internal_reduce_allocate(c))
      return rc;
  }
  if (r.pixels != 0u) {
    HIP_OK(c, hipMemsetAsync(r.snapshot, 0, r.pixels * kFilmLayers * sizeof(float4), r.stream));
    HIP_OK(c, hipMemsetAsync(r.reduced, 0, r.pixels * kFilmLayers * sizeof(float4), r.stream));
    HIP_OK(c, hipStreamSynchronize(r.stream));
  }
  return ETX_HIP_OK;
}

// First half of a reduce, everything before the collectives: the snapshot of the film behind every lane's newest commit, and this rank's
// counter words {iterations it counts, 1 if it failed}. Never waits for the lanes.
// Which layers: what the armed integrator writes (VCM: camera + light; path tracer: camera + normal + albedo; bidirectional: all four). The
// normal / albedo sums of the path tracer and the bidirectional integrator are added by the shade kernels of iterations in flight, not by
// their commits: a snapshot taken while lanes render may hold a part of those iterations' AOV values (progressive display only; after
// etx_hip_sync - which etx_hip_reduce_film does - every layer holds whole iterations).
int etx_hip_internal_reduce_prepare(etx_hip_context* c, int local_rc, float4** out_snapshot, float4** out_reduced, size_t* out_pixels, uint32_t* out_layer_mask) {
  EtxReduceState& r = c->reduce;
  if (c->scene_ready == false) {
    c->error = "film reduce: no scene uploaded";
    return ETX_HIP_ERROR_STATE;
  }
  if (int rc = etx_hip_internal_reduce_allocate(c))
    return rc;
  const size_t n = r.pixels;
  const uint32_t layer_mask = (c->integrator == ETX_HIP_INTEGRATOR_VCM) ? 0x3u : ((c->integrator == ETX_HIP_INTEGRATOR_PT) ? 0xdu : 0xfu);
  // the bidirectional integrator's commit counts every pixel of the frame on every pixel shard (k_vcm_commit): each shard contributes the counts of the
  // pixels it owns (k_film_snapshot)
  const uint32_t own_stride = (c->integrator == ETX_HIP_INTEGRATOR_BDPT) ? c->pixel_stride : 1u;
  uint32_t counted = 0;
  {
    std::lock_guard<std::mutex> lock(c->shared_mutex);
    // ranks that share their iterations and split the PIXELS hold the same iterations: the rank of the first pixel shard counts them
    counted = (c->pixel_first == 0u) ? c->local_iterations : 0u;
    if ((local_rc == 0) && (c->sticky_error != 0))
      local_rc = c->sticky_error;  // an iteration in flight has failed since the last call
  }
  {
    std::lock_guard<std::mutex> lock(r.mutex);
    if (c->commit_recorded)
      HIP_OK(c, hipStreamWaitEvent(r.stream, c->commit_done, 0));
    for (etx_hip_context* helper : c->helpers) {
      if (helper->commit_recorded)
        HIP_OK(c, hipStreamWaitEvent(r.stream, helper->commit_done, 0));
    }
    if (c->read_pending && (c->read_event != nullptr))
      HIP_OK(c, hipStreamWaitEvent(r.stream, c->read_event, 0));  // an asynchronous read-back of the reduced copy still in flight
    HIP_OK(c, hipEventRecord(r.time_begin, r.stream));
    launch_film_snapshot(r.stream, c->pipe.camera_sum, r.snapshot, uint32_t(n), layer_mask, c->pixel_first, own_stride, c->scene.film_w, c->scene.film_h);
    HIP_OK(c, hipEventRecord(r.snapshot_done, r.stream));
    r.snapshot_recorded = true;
  }
  launch_set_words(r.stream, r.d_counters, counted, local_rc ? 1ull : 0ull);  // by value: several reduces may be in flight
  r.layer_mask = layer_mask;
  r.payload_bytes = uint64_t(__builtin_popcount(layer_mask)) * n * sizeof(float4);
  *out_snapshot = r.snapshot, *out_reduced = r.reduced, *out_pixels = n, *out_layer_mask = layer_mask;
  return ETX_HIP_OK;
}

// etx_hip_internal_reduce_prepare failed on this rank although the communicator and the buffers exist (a HIP call inside it): the rank still joins the
// collective the others are entering - with a zero snapshot and its failed flag set - so that every rank returns an error instead of waiting for the
// RCCL timeout. Non-zero: there is nothing to join with (no scene, no buffers: the documented case of etx_hip.h).
int etx_hip_internal_reduce_prepare_failed(etx_hip_context* c, float4** out_snapshot, float4** out_reduced, size_t* out_pixels, uint32_t* out_layer_mask) {
  EtxReduceState& r = c->reduce;
  if ((r.stream == nullptr) || (r.d_counters == nullptr) || (r.snapshot == nullptr) || (r.reduced == nullptr) || (r.pixels == 0u))
    return ETX_HIP_ERROR_STATE;
  const uint32_t layer_mask = (c->integrator == ETX_HIP_INTEGRATOR_VCM) ? 0x3u : ((c->integrator == ETX_HIP_INTEGRATOR_PT) ? 0xdu : 0xfu);
  if (hipMemsetAsync(r.snapshot, 0, r.pixels * kFilmLayers * sizeof(float4), r.stream) != hipSuccess)
    return ETX_HIP_ERROR_HIP;
  launch_set_words(r.stream, r.d_counters, 0ull, 1ull);
  r.layer_mask = layer_mask;
  *out_snapshot = r.snapshot, *out_reduced = r.reduced, *out_pixels = r.pixels, *out_layer_mask = layer_mask;
  return ETX_HIP_OK;
}

// Second half, behind the collectives: the received counter words travel to pinned memory, the reduce's end is recorded.
int etx_hip_internal_reduce_finish(etx_hip_context* c) {
  EtxReduceState& r = c->reduce;
  HIP_OK(c, hipMemcpyAsync(r.h_counters + 2, r.d_counters + 2, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, r.stream));
  HIP_OK(c, hipEventRecord(r.time_end, r.stream));
  HIP_OK(c, hipEventRecord(r.done, r.stream));
  return ETX_HIP_OK;
}
