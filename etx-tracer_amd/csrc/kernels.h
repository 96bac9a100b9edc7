// kernels.h - host-callable launch wrappers of the gfx950 kernels (definitions in kernels_*.hip).
#pragma once

#include <hip/hip_runtime.h>
#include "pipeline.h"

namespace etxd {

constexpr uint32_t kBlockSize = 256;
constexpr uint32_t kPersistentBlocks = 256 * 8;  // 256 CUs x 8 blocks of 256 threads, grid-stride loops

// Workgroups of kBlockSize threads of `kernel` the device holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units, cached per
// kernel). The persistent kernels grid-stride over their items: more workgroups than that only queue behind the resident ones and pay their
// prologue again. Applied to the simple group's shade kernels (two workgroups per CU: 512 instead of 2048, 105.4 -> 106.6 Msamples/s on configs[1]); the
// same cap on the pair, merge and shadow kernels changed nothing (profiles/round4_ab_lds_tables_and_grid.txt) and is not applied.
uint32_t resident_blocks(const void* kernel);
// grid of a persistent kernel: what the items need, at most `percent` % of the resident workgroups (0: at most kPersistentBlocks)
uint32_t persistent_grid(uint32_t blocks_needed, const void* kernel, uint32_t percent);

// Shading groups the scene holds besides "simple" (dev_scene.h kShadeGroup*, set at upload from the materials in use)
struct ShadeGroups {
  bool general = false, subsurface = false;
  bool binned() const {
    return general || subsurface;
  }
  uint32_t widest() const {  // the group whose kernel shades every material of the scene (tail kernels)
    return subsurface ? uint32_t(kShadeGroupSubsurface) : (general ? uint32_t(kShadeGroupGeneral) : uint32_t(kShadeGroupSimple));
  }
};

// traversal
// `max_items`: host-side upper bound of the device-resident item count (sizes the grid; kernels grid-stride anyway)
constexpr uint32_t kRoundMirrorSlots = 1024;  // ring of (round tag, active count) entries the trace kernel writes to pinned host memory
void launch_trace_closest(hipStream_t stream, const Pipeline& p, uint32_t set, uint32_t active_counter, uint32_t max_items, bool flat, unsigned long long* round_mirror, uint32_t round_tag, uint32_t pass_stat,
  uint32_t cross_mode);  // pass_stat: kStatRaysLight / kStatRaysCamera / 0; cross_mode: 0 / 1 (VCM state) / 2 (bidirectional state): medium boundaries crossed inside the kernel (kernels_trace.hip kCross)
void launch_trace_shadow(hipStream_t stream, const Pipeline& p, uint32_t max_items, bool flat);
void launch_trace_rays(hipStream_t stream, const DScene& scene, const float4* ray_o_tmin, const float4* ray_d_tmax, float4* hits, uint32_t count, bool flat, uint32_t debug_flags);

// VCM light pass
void launch_stats_finalize(hipStream_t stream, const Pipeline& p);
void launch_iteration_reset(hipStream_t stream, const Pipeline& p);
void launch_light_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it);
void launch_light_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups);

// path tracer (kernels_pt.hip)
void launch_pt_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it);
void launch_pt_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups);
void launch_pt_commit(hipStream_t stream, float4* iteration_image, float4* camera_sum, float4* adaptive_sum, uint32_t pixels, float radiance_clamp);
void launch_noise_estimate(hipStream_t stream, const Pipeline& p, uint32_t width, uint32_t height, float threshold);  // Film::estimate_noise_levels

// tail: the few paths that are still alive after many bounces finish inside one launch (kernels_tail.hip)
void launch_light_tail(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups);
void launch_camera_tail(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups);

// bidirectional path tracing (kernels_bdpt.hip)
// which BSDF instantiations a pass launches: DeviceScene::simple_materials / ::bdpt_binning (kernels_bdpt.hip ETX_BDPT_LAUNCH)
enum : uint32_t { kBdptKernelsGeneral = 0u, kBdptKernelsSimple = 1u, kBdptKernelsBinned = 2u };
void launch_bdpt_light_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it);
void launch_bdpt_light_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, uint32_t variant);  // variant: kBdptKernels*
void launch_bdpt_walk(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool camera, uint32_t in_set, uint32_t max_items, uint32_t variant);  // subsurface walks of the round: walk queue in_set -> path set / walk queue in_set ^ 1
void launch_bdpt_connect_camera(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items, uint32_t variant);
void launch_bdpt_camera_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it);
void launch_bdpt_camera_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, uint32_t variant);
void launch_bdpt_connect_light(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items, uint32_t variant);
void launch_bdpt_connect_pairs(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items, uint32_t variant);
void launch_bdpt_expand_pairs(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t max_items);

// photon grid
void launch_grid_build(hipStream_t stream, const Pipeline& p, const VcmParams& it);

// VCM camera pass
void launch_camera_generate(hipStream_t stream, const Pipeline& p, const VcmParams& it);
void launch_camera_shade(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t max_items, const ShadeGroups& groups);
void launch_connect_endpoints(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool camera_pass, uint32_t max_items);  // general / subsurface groups (pipeline.h EndpointQueue)
void launch_connect(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool generic_materials, uint32_t max_items);
void launch_merge_reset(hipStream_t stream, const Pipeline& p);
void launch_merge(hipStream_t stream, const Pipeline& p, const VcmParams& it, bool generic_materials, uint32_t max_items);

// film
void launch_vcm_commit(hipStream_t stream, float4* iteration_camera, float4* iteration_light, float4* camera_sum, float4* light_sum, uint32_t pixels, const uint32_t* counters);
void launch_set_words(hipStream_t stream, unsigned long long* dst, unsigned long long a, unsigned long long b);  // the counter words of a film reduce
void launch_film_snapshot(hipStream_t stream, const float4* film, float4* snapshot, uint32_t pixels, uint32_t layer_mask, uint32_t own_first, uint32_t own_stride, uint32_t film_w, uint32_t film_h);  // multi-GPU reduce, host_reduce.h
void launch_film_resolve(hipStream_t stream, const float4* camera_sum, const float4* light_sum, float4* out, uint32_t pixel_count, float scale, int layer, const float4* counts = nullptr);

// known-answer kernels
void launch_kat(hipStream_t stream, int which, const float* in, uint32_t count, float* out, const uint2* bluenoise);

}  // namespace etxd
