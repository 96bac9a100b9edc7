/*
 * etx_hip.h - C ABI of libetx_hip.so, the MI355X (gfx950) wavefront light-transport backend for etx-tracer.
 *
 * Drop-in boundary: the reference's plugin interface is `struct Integrator`
 * (sources/etx/rt/integrators/integrator.hxx:12-98). A host class `HIPVCM : Integrator` forwards
 *   run()     -> etx_hip_upload_scene + etx_hip_begin + etx_hip_render_iteration   (replaces vcm_cpu.cxx:81-124:
 *                Film::clear, VCMOptions::load, scheduler.schedule(pixel_count, &light_gather))
 *   update()  -> etx_hip_poll / next etx_hip_render_iteration / etx_hip_read_film   (replaces vcm_cpu.cxx:264-276
 *                + complete_light_vertices :209-225 + complete_camera_vertices :227-241)
 *   stop()    -> etx_hip_sync                                                       (replaces vcm_cpu.cxx:278-288)
 *   status()  -> etx_hip_stats                                                      (Integrator::Status, integrator.hxx:24-37)
 * i.e. the device boundary sits exactly where the reference calls TaskScheduler::schedule for its per-pixel loops.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success, <0 on error (etx_hip_last_error
 * gives the text), mirroring the reference's bool + log::error convention (sources/etx/gpu/gpu.hxx:74).
 * The library never falls back to a CPU path: without a gfx950 device etx_hip_create fails.
 */
#ifndef ETX_HIP_H
#define ETX_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "etx_scene_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ETX_HIP_ABI_VERSION 4 /* 3: reference_seeding in the option structs, the asynchronous film reduce (etx_hip_reduce_film_begin / _end / _info);
                                 4: etx_hip_runtime_info, etx_hip_comm_all_reduce_f64 / _barrier, etx_hip_trace_rays_timed */

typedef struct etx_hip_context etx_hip_context; /* opaque */

enum {
  ETX_HIP_OK = 0,
  ETX_HIP_ERROR_INVALID_ARGUMENT = -1,
  ETX_HIP_ERROR_NO_DEVICE = -2,
  ETX_HIP_ERROR_HIP = -3,            /* a HIP runtime call failed */
  ETX_HIP_ERROR_UNSUPPORTED = -4,    /* scene feature not implemented by the device path (never silently ignored) */
  ETX_HIP_ERROR_STATE = -5,          /* call order violated (e.g. render before begin) */
  ETX_HIP_ERROR_OVERFLOW = -6,       /* a device queue/pool overflowed during the last iteration */
  ETX_HIP_ERROR_COMM = -7            /* RCCL failure */
};

/* which integrator the device pipeline runs; names follow the reference classes it stands in for */
enum {
  ETX_HIP_INTEGRATOR_PT = 0,   /* CPUPathTracing  sources/etx/rt/integrators/path_tracing.cxx:50-110, options = etx_abi_pt_options */
  ETX_HIP_INTEGRATOR_VCM = 1,  /* CPUVCM          sources/etx/rt/integrators/vcm_cpu.cxx:95-241,     options = etx_abi_vcm_options */
  ETX_HIP_INTEGRATOR_BDPT = 2  /* CPUBidirectional sources/etx/rt/integrators/bidirectional.cxx:342-403, options = etx_abi_bdpt_options
                                  (all four modes, random-walk subsurface materials included: their walks run as their own kernels) */
};

/* film layers, subset of etx::Film layer ids (sources/etx/render/host/film.hxx:14-27) that the MC loop produces */
enum {
  ETX_HIP_LAYER_CAMERA = 0, /* Film::CameraImage : running mean of per-iteration camera estimates */
  ETX_HIP_LAYER_LIGHT = 1,  /* Film::LightImage  : running mean of light-path splats */
  ETX_HIP_LAYER_RESULT = 2, /* Film::Result      : max(0, camera + light), film.cxx:401-409 */
  ETX_HIP_LAYER_NORMAL = 3, /* Film::Normals     : running mean of the first-hit shading normal, as Film::layer returns it: n * 0.5 + 0.5 (PT), film.cxx:207-211,411 */
  ETX_HIP_LAYER_ALBEDO = 4  /* Film::Albedo      : running mean of the first-hit bsdf::albedo (PT), film.cxx:212-216 */
};

/* ------------------------------------------------------------------------------------------------------------ */
/* lifetime */

int etx_hip_abi_version(void);

/* Opens HIP device `device` (must be gfx950) and creates the streams / queues.
 * Environment the library reads (deployment settings, nothing else is read from the environment by the product build):
 *   ETX_HIP_LANES=n                      iterations in flight, 1..8 (default: four, six for the bidirectional integrator) - etx_hip_create
 *   ETX_HIP_BVH_BUILD_THREADS=n          host threads of the binned-SAH tree build (default: the hardware's) - etx_hip_upload_scene
 *   ETX_HIP_VERBOSE=1                    build phases and their times on stderr */
int etx_hip_create(int device, etx_hip_context** out_context);
void etx_hip_destroy(etx_hip_context* context);

/* Text of the last error on this context (or of the last failed etx_hip_create when context == NULL). */
const char* etx_hip_last_error(const etx_hip_context* context);

/* ------------------------------------------------------------------------------------------------------------ */
/* scene: replaces Raytracing::commit_changes (sources/etx/rt/rt.cxx:58-88: film.allocate + Embree scene build)
 * and the author's disabled build_device_scene sketch (rt.cxx:141-238). Borrows the host arrays during the call,
 * owns device copies afterwards: geometry, BVH (built here), materials with spectra pre-resolved for the scene's
 * RGB/spectral mode, emitters + distribution, images + sampling tables, media. Allocates the film for
 * camera->film_size. */
int etx_hip_upload_scene(etx_hip_context* context, const etx_abi_scene* scene, const etx_abi_camera* camera);

/* Scene edits without a full upload. The reference re-commits the whole scene on every change - Embree rebuilds its
 * BVH in Raytracing::commit_changes (rt.cxx:58-88) whenever the camera, a material or the geometry was touched (app.cxx:368-399).
 * Here the caller says what it edited in the scene it uploaded before:
 *   ETX_HIP_CHANGED_CAMERA    camera parameters (same film size)
 *   ETX_HIP_CHANGED_MATERIALS materials, spectra, emitter profiles / instances / distribution, medium parameters, scene scalars
 *                             (samples, path lengths, clamp, noise threshold); the traversal filters (Void, Boundary, alpha test)
 *                             follow the new material classes
 *   ETX_HIP_CHANGED_POSITIONS vertex data moved (same vertex and triangle counts, same indices): the vertices and triangles are
 *                             copied over, the traversal triangles re-derived and the BVH boxes refit bottom-up ON THE DEVICE
 *                             (kernels_bvh_build.hip); the tree keeps its topology, so a refit after large deformations traverses
 *                             slower than a rebuild
 *   ETX_HIP_REBUILD_BVH       with ETX_HIP_CHANGED_POSITIONS: instead of the refit, a new tree is built over the moved vertices ON THE
 *                             DEVICE (linear BVH, see etx_hip_set_bvh_builder) - for deformations a refit tree traverses badly
 * The small tables and the camera are rebuilt from `scene` / `camera` in any case; vertices, triangles, BVH, image pixels and
 * density grids stay resident. Counts of vertices, triangles, images and media must be unchanged (ETX_HIP_ERROR_INVALID_ARGUMENT
 * otherwise; on any error the context holds no scene, as after a failed upload). Waits for the iterations in flight; the next
 * etx_hip_begin renders the edited scene. */
enum { ETX_HIP_CHANGED_CAMERA = 1, ETX_HIP_CHANGED_MATERIALS = 2, ETX_HIP_CHANGED_POSITIONS = 4, ETX_HIP_REBUILD_BVH = 8 };
int etx_hip_update_scene(etx_hip_context* context, const etx_abi_scene* scene, const etx_abi_camera* camera, uint32_t changed);

/* Who builds the traversal tree of the NEXT etx_hip_upload_scene (Raytracing::commit_changes hands this to Embree, rt.cxx:66-88):
 *   ETX_HIP_BVH_HOST_SAH     (default) binned-SAH BVH2 on the host, collapsed to the four-wide nodes: the tree that traverses fastest
 *   ETX_HIP_BVH_DEVICE_LBVH  linear BVH on the device (dev_lbvh.h: 63-bit Morton keys, radix sort, Karras' binary radix tree,
 *                            surface-area guided collapse to four-wide breadth-first nodes, boxes bottom-up): a few milliseconds
 *                            for 10^6 triangles - time to first iteration, geometry that changes every frame - at 3-15 % of the
 *                            traversal rate. Scenes of <= 64 triangles are swept linearly and always built on the host.
 * (An eight-wide tree with 8-bit child boxes, ETX_HIP_BVH_WIDE = 256 until ABI 2, was measured no faster than the four-wide tree on any workload -
 * lower lane utilisation, more VALU per ray - and is gone: DESIGN.md 3, HISTORY.md.)
 * etx_hip_bvh_info: {BVH4 nodes, triangles, depth | traversal stack entries << 16, bytes} of the uploaded scene and the time its
 * tree took to build (host: wall clock of the builder; device: HIP events around the build kernels), in milliseconds. */
enum { ETX_HIP_BVH_HOST_SAH = 0, ETX_HIP_BVH_DEVICE_LBVH = 1 };
int etx_hip_set_bvh_builder(etx_hip_context* context, int builder);
int etx_hip_bvh_info(etx_hip_context* context, uint32_t out_info[4], double* out_build_ms);

/* The blue-noise samples options.blue_noise needs (vcm_shared.hxx:941-945, 1018-1022). The host's sampler is
 * sample_blue_noise(pixel, scene.samples, iteration, dimension) (path_tracing.cxx:173-178 -> thirdparty/bluenoise
 * BNSampler), whose tables are private to that library; the ABI therefore takes the sampler's OUTPUT for one
 * sample-count class, tabulated by the host:
 *   values[(((py * 128 + px) * 256) + sample) * 8 + dimension] = (uint8_t)(BNSampler(px, py, samples, sample).get(dimension) * 256)
 * for px, py < 128, sample < 256, dimension < 8 (the sampler wraps exactly these ranges; its floats are
 * (0.5 + value) / 256). `set_index` = log2 of the class (0..8: 1, 2, 4, ... 256 samples; BNSampler picks
 * next_pow2(clamp(scene.samples, 1, 256))). `bytes` must be 128*128*256*8. etx_hip_begin fails with
 * ETX_HIP_ERROR_UNSUPPORTED when options.blue_noise is set and the class of scene.samples has not been uploaded. */
int etx_hip_upload_bluenoise(etx_hip_context* context, uint32_t set_index, const uint8_t* values, size_t bytes);

/* Spectral scenes (Scene::spectral()): the film conversion (value / sampling_pdf).to_rgb() needs the CIE observer the
 * host compiles in, spectrum::spectral_xyz(i) for i < spectrum::WavelengthCount at 1 nm from spectrum::kShortestWavelength
 * (sources/etx/render/shared/spectrum.hxx:17-21, 24-186). xyz = count * 3 floats. etx_hip_begin fails with
 * ETX_HIP_ERROR_UNSUPPORTED on a spectral scene until the table is uploaded. */
int etx_hip_upload_cie_table(etx_hip_context* context, const float* xyz, uint32_t count, float first_wavelength);

/* Spectral scenes with RGB textures (albedo / emission images, image environment maps): apply_rgb
 * (sources/etx/render/shared/scene.hxx:249-260) weighs the texel with rgb_response(wavelength, rgb), a table the host compiles
 * in (sources/etx/render/host/spectrum.cxx:399-612, 1 nm steps from spectrum::kRGBResponseShortestWavelength). rgb = count * 3
 * floats: rgb_response of the unit colours (1,0,0), (0,1,0), (0,0,1) at every integer wavelength = the rows of that table.
 * etx_hip_begin fails with ETX_HIP_ERROR_UNSUPPORTED on such a scene until the table is uploaded. */
int etx_hip_upload_rgb_response(etx_hip_context* context, const float* rgb, uint32_t count, float first_wavelength);

/* ------------------------------------------------------------------------------------------------------------ */
/* rendering */

/* Clears the film (Film::clear(ClearCameraData|ClearLightData), vcm_cpu.cxx:86) and arms the pipeline.
 * `options`: etx_abi_vcm_options, etx_abi_pt_options or etx_abi_bdpt_options (by integrator). Scene scalars (samples, min/max path length,
 * random_path_termination, radiance_clamp) come from the uploaded scene.
 * This context renders iterations first_iteration, first_iteration + iteration_stride, ... (multi-GPU sharding by
 * iteration; single GPU: 0, 1).
 * Path tracing with Scene::noise_threshold > 0 (adaptive sampling, Film::estimate_noise_levels): iterations run one after the other
 * on ONE device lane, each reading the convergence mask its predecessor left (as the reference does, path_tracing.cxx:91-99), so the
 * render is reproducible; on an iteration-sharded context (iteration_stride != 1) the mask would be a property of a film no rank
 * holds, and every pixel is sampled in every iteration instead (noise_threshold treated as 0). */
int etx_hip_begin(etx_hip_context* context, int integrator, const void* options, size_t options_size, uint32_t first_iteration, uint32_t iteration_stride);

/* etx_hip_begin with a second way to shard a run ("tile-split"): this context renders pixels pixel_first, pixel_first +
 * pixel_stride, ... of every one of its iterations (pixels in the reference's order, y * width + x). For the path tracer and the
 * bidirectional integrator only: both pair light path i with pixel i (bidirectional.cxx:379-391), so a context traces the emitter and camera
 * paths of ITS pixels - the same samples as an unsharded render, seeded by (pixel, iteration). The camera image of the context is tile-local
 * (other pixels stay zero), its light image is full-frame (splats land anywhere, bidirectional.cxx:516-520) and holds the splats of its own
 * light paths; the job's film is the SUM of the contexts' films - the same single all-reduce of zero-padded sums as iteration sharding
 * (etx_hip_reduce_film; the rank of pixel shard 0 contributes the iteration count), or the sum of the etx_hip_read_film images. The
 * growable pools start at the share of the paths (half of everything for pixel_stride 2). Adaptive sampling is off on a pixel-sharded
 * context (the mask reads row and column neighbours, film.cxx:283-321). VCM: ETX_HIP_ERROR_UNSUPPORTED unless (0, 1) - the photon map
 * of an iteration needs the light paths of every pixel. etx_hip_begin = (0, 1). */
int etx_hip_begin_ex(etx_hip_context* context, int integrator, const void* options, size_t options_size, uint32_t first_iteration, uint32_t iteration_stride, uint32_t pixel_first,
  uint32_t pixel_stride);

/* Hands one full iteration (VCM: light pass, grid build, camera pass; PT: one sample per pixel) to a free device lane and
 * returns without waiting for it; blocks only while every lane is busy (four for VCM and path tracing, six for the
 * bidirectional integrator; ETX_HIP_LANES=n, 1..8, fixes one count for all three). */
int etx_hip_render_iteration(etx_hip_context* context);

/* The same without ever blocking: 1 = the iteration was handed to a free device lane, 0 = every lane is busy (call again
 * later, e.g. at the next Integrator::update()), <0 = error. Integrator::update must not block (vcm_cpu.cxx:264-268: CPUVCM
 * returns at once while its task set is incomplete); etx_hip_render_iteration blocks while every lane is busy. */
int etx_hip_try_render_iteration(etx_hip_context* context);

/* 1 = all enqueued iterations finished, 0 = still running, <0 = error. Never blocks (Integrator::update must not
 * block: vcm_cpu.cxx:264-268). */
int etx_hip_poll(etx_hip_context* context);

/* Blocks until the device is idle (Integrator::stop(Immediate), vcm_cpu.cxx:278-288). */
int etx_hip_sync(etx_hip_context* context);

/* Copies a film layer as float4 RGBA (alpha = 1), row order and y-flip as etx::Film stores it
 * (film.cxx:165,189: row (H-1-y)). A context without a communicator: waits for the iterations in flight, the image is this context's own
 * film normalised by its iterations. A context WITH a communicator that has finished a reduce in this run: the reduced copy (whole job, see
 * etx_hip_reduce_film_*; does not wait for iterations in flight, does wait for a reduce in flight). */
int etx_hip_read_film(etx_hip_context* context, int layer, float* dst_rgba, size_t dst_bytes);

/* Asynchronous read-back: _begin enqueues the resolve and the device-to-host copy of `layer` on a stream of
 * its own and returns at once, without waiting for iterations in flight (the sums hold completed iterations only, so the
 * snapshot is a valid progressive image normalised by the iterations completed so far). _end copies the image out of the
 * pinned staging buffer: wait = 0 never blocks (returns 0 while the copy is still running, 1 when dst has been filled),
 * wait = 1 blocks until it has arrived. One read-back can be pending per context. */
int etx_hip_read_film_begin(etx_hip_context* context, int layer);
int etx_hip_read_film_end(etx_hip_context* context, float* dst_rgba, size_t dst_bytes, int wait);

/* Checkpoint / resume (the reference has none: a stopped render starts over, app.cxx:193-216). A checkpoint is the
 * film state of this context - the sums of its completed iterations with their per-pixel sample counts, the adaptive-sampling state,
 * and the index of the next iteration - as one host buffer of etx_hip_checkpoint_bytes() bytes (0 before etx_hip_begin), written
 * by _save (waits for the iterations in flight) any time after etx_hip_begin (film reduces are out of place: the sums stay this rank's own). _load, called after an etx_hip_begin
 * with the same integrator, options, film size and iteration sharding (checked, ETX_HIP_ERROR_INVALID_ARGUMENT otherwise), replaces
 * the film with the saved one and continues at the saved iteration: iterations are seeded by (pixel, iteration), so the
 * resumed render is the render that was interrupted. The buffer is the caller's to store (file, object store). */
size_t etx_hip_checkpoint_bytes(const etx_hip_context* context);
int etx_hip_checkpoint_save(etx_hip_context* context, void* dst, size_t dst_bytes);
int etx_hip_checkpoint_load(etx_hip_context* context, const void* src, size_t src_bytes);

typedef struct etx_hip_stats_t {
  /* Integrator::Status (integrator.hxx:24-37) */
  double last_iteration_time; /* seconds, device time of the last finished iteration (HIP events) */
  double total_time;
  uint32_t completed_iterations;
  uint32_t current_iteration;
  /* counters, TOTALS since etx_hip_begin ("counters the oracle must emit"); divide by completed_iterations for
   * per-iteration figures */
  uint64_t rays_extension;       /* closest-hit rays (light + camera sub paths) */
  uint64_t rays_shadow;          /* transmittance rays (camera connections, NEE, vertex connections) */
  uint64_t light_vertices;       /* stored light vertices */
  uint64_t camera_vertices;      /* connectible camera vertices */
  uint64_t photons_examined;     /* photons visited by the merge */
  uint64_t photons_merged;       /* photons accepted by the merge */
  uint64_t splats;               /* light image splats */
  uint64_t wavefront_bounces;    /* kernel rounds (light + camera) */
  uint32_t overflow_flags;       /* != 0: a pool overflowed and could not grow (that iteration failed with ETX_HIP_ERROR_OVERFLOW and is not in the film) */
  uint32_t nonfinite_dropped;    /* film contributions that were not finite and were dropped instead of poisoning their pixel (expected: 0) */
  /* per-kernel device time since etx_hip_begin, milliseconds (HIP events on the launch streams, summed over the iterations) */
  double ms_trace_closest;
  double ms_trace_shadow;
  double ms_shade_light;
  double ms_shade_camera;
  double ms_connect;
  double ms_merge;
  double ms_grid_build;
  double ms_generate;
  uint64_t launches_trace_closest;
  uint64_t launches_trace_shadow;
  /* more totals since etx_hip_begin (the units the per-kernel rooflines of bench.py are computed from) */
  uint64_t rays_light;           /* closest-hit rays of the light pass */
  uint64_t rays_camera;          /* closest-hit rays of the camera pass (PT: all rays) */
  uint64_t pairs;                /* (camera vertex, light vertex) pairs listed for connection (bidirectional integrator, since round 6: without the emitter's own vertex, which connects to nothing) */
  uint64_t endpoints;            /* endpoint connections of the general / subsurface shading groups (k_connect_endpoints) */
  /* adaptive sampling (path tracing with Scene::noise_threshold > 0: Film::estimate_noise_levels / active_pixel) */
  uint64_t active_pixels;        /* pixels sampled, total since etx_hip_begin */
  uint64_t last_active_pixels;   /* pixels sampled by the most recently finished iteration: 0 = every pixel has converged (CPUPathTracing stops, path_tracing.cxx:91-93) */
  uint64_t boundary_crossings;   /* closest-hit queries the traversal kernel ran beyond a medium boundary it crossed itself (paths outside any medium:
                                    vcm_handle_boundary_bsdf draws nothing); included in rays_extension, not in rays_light / rays_camera (segments shaded) */
  /* ABI 2 */
  uint32_t pool_grows;           /* times an iteration overflowed a device pool, was discarded before its commit and rendered again with larger pools */
  uint32_t reserved_0;
} etx_hip_stats_t;

/* Which kernel groups are timed with HIP events (bit i = the i-th ms_* field above, in declaration order; default: the two
 * traversal groups). Timing costs two events per launch; bench.py switches everything on for its per-kernel roofline pass. */
int etx_hip_set_timers(etx_hip_context* context, uint32_t mask);

int etx_hip_stats(etx_hip_context* context, etx_hip_stats_t* out_stats, size_t stats_size);

/* Ablation switches of the kernels for timing experiments and kernel-level tests (0 = production, the default): bit 0 no film
 * atomics in the shadow kernel, bit 2 no transmittance traversal, bit 6 (64) the packed two-ray flat sweep in etx_hip_trace_rays*,
 * bit 7 (128) the matrix-core flat sweep, bits 8 / 9 no next event estimation / no camera vertex storage, bit 16 (0x10000) the bidirectional kernels of a
 * scene that mixes Lambert surfaces with materials of the general BSDF classes run their general instantiation for every item instead of splitting the items
 * by class. (The reference's seeding of the camera path, bit 15 until ABI 2, is a product
 * option now: etx_abi_vcm_options / etx_abi_bdpt_options::reference_seeding.) Waits for the iterations in flight. The library reads no
 * environment variable for these; builds with -DETX_HIP_DEBUG additionally read tuning knobs (csrc/tuning_knobs.h). */
int etx_hip_set_debug_flags(etx_hip_context* context, uint32_t flags);

/* The device pools whose fill depends on the scene and on the sample (light vertices and the photon grid built over them, camera vertices,
 * connection pairs, shadow and endpoint queues) start at what typical paths need - five stored light vertices per path, sixteen in scenes
 * with subsurface materials - and grow on demand: an iteration that overflows a pool is discarded before it reaches the film, the pool is
 * doubled for all lanes and the same iteration is rendered again (etx_hip_stats_t::pool_grows counts these). The reference grows a
 * std::vector under a mutex at the same place (vcm_cpu.cxx:131-171). This call sets, for the NEXT etx_hip_upload_scene / _update_scene:
 * the light vertices per path the pools start with (0 = the default) and the bytes one lane's pools may grow to (0 = no limit but the
 * device's memory); beyond the limit an overflow fails the iteration with ETX_HIP_ERROR_OVERFLOW, as a failed allocation would. */
int etx_hip_set_pool_policy(etx_hip_context* context, uint32_t initial_light_vertices_per_path, size_t max_pool_bytes_per_lane);

/* Iterations the context keeps in flight (device lanes) under the given integrator: what ETX_HIP_LANES set, else four, six for the
 * bidirectional integrator. */
uint32_t etx_hip_lanes(const etx_hip_context* context, int integrator);

/* Device memory of the per-iteration working sets (queues, vertex pools, photon grid, film) of all lanes, in bytes: what a render of
 * the uploaded scene with the integrators used so far holds besides the scene itself. The photon grid of a lane exists from the first
 * etx_hip_begin(VCM) on, the pools of the fifth and sixth lane from the first etx_hip_begin(BDPT) on; both until the next
 * etx_hip_upload_scene / etx_hip_update_scene. */
size_t etx_hip_device_bytes(const etx_hip_context* context);

/* ------------------------------------------------------------------------------------------------------------ */
/* multi GPU: iterations (etx_hip_begin first / stride) or pixels (etx_hip_begin_ex) are sharded over ranks; the only exchange is an RCCL
 * sum-reduce of the float4 film layers the armed integrator writes - VCM: camera + light, 32 B per pixel (66 MB at 1080p); path tracer: camera +
 * normal + albedo; bidirectional: all four - plus two counter words. The 128-byte ncclUniqueId is created by rank 0 with
 * etx_hip_comm_unique_id and distributed by the host (etx_tracer_amd/multi_gpu.py: a file on the node, or the host's own process group). */
#define ETX_HIP_UNIQUE_ID_BYTES 128
int etx_hip_comm_unique_id(void* out_id_128_bytes);
/* Joins the communicator and allocates what a reduce needs (communication stream, events, counters; the two film-sized buffers as soon as a scene
 * is uploaded - here or at the next etx_hip_begin), so that a rank finds out HERE, not inside a collective, that it cannot take part. */
int etx_hip_comm_init(etx_hip_context* context, int rank, int world_size, const void* id_128_bytes);

/* The film reduce. The reference's film is consumed while the render runs (Film::commit_light_iteration once per iteration, film.cxx:332-343; the
 * GUI reads it every frame, app.cxx:150-155) and north_star places the reduce "at the end of each iteration": a reduce therefore neither ends the
 * render nor waits for it. Every rank calls the same sequence of reduces (they are collectives); any cadence - after every iteration, every k-th,
 * once at the end - gives the same final film.
 *
 *   etx_hip_reduce_film_begin  enqueues, on a communication stream of the context's own: (1) a SNAPSHOT of the film sums behind every commit the
 *       lanes have enqueued so far - commits and snapshots exclude each other on the device, so a snapshot holds whole iterations; the lanes keep
 *       rendering and no host thread waits; (2) ncclAllReduce(sum) OUT OF PLACE from the snapshot into the context's reduced copy, and of the
 *       counter words {iterations, failed flag}. Returns at once - it waits neither for the lanes nor for earlier reduces: any number may be in
 *       flight, the communication stream orders them (a reduce per iteration costs the rendering nothing but the device time of a 66 MB copy and
 *       the collective). Without a communicator (single GPU): nothing to do, returns ETX_HIP_OK.
 *   etx_hip_reduce_film_end    collects every reduce begun so far. wait = 0: 0 while the newest is running, 1 once the reduced copy is complete (also when
 *       none was in flight); wait = 1: blocks until it is. < 0: error - this rank's own failed iteration, ETX_HIP_ERROR_COMM when another rank reported one (every
 *       rank always joins the collective; the error travels with the counters so that nobody waits for the RCCL timeout), or a HIP / RCCL error.
 *   etx_hip_reduce_film        = etx_hip_sync + _begin + _end(wait): every iteration handed over so far, on every rank, is in the reduced copy
 *       when it returns. Rendering may continue afterwards (until ABI 2 the reduce was in place and final).
 * After the first finished reduce of a run, etx_hip_read_film / _read_film_begin on a context WITH a communicator return the reduced copy - the
 * whole job's film as of the newest finished reduce, normalised pixel by pixel by the reduced sample count (the camera layer's w), so ranks need
 * not have completed the same number of iterations; a context without a communicator keeps returning its own film. The film sums of the rank
 * itself are never modified: checkpoints, statistics and further iterations are unaffected. etx_hip_begin starts a new run (a reduce still in
 * flight is waited for, the reduced copy is cleared). Normal / albedo values of the path tracer and the bidirectional integrator are added by the
 * shade kernels of iterations in flight: a reduce taken while lanes render can hold part of those iterations' AOV values (a progressive display
 * issue only; etx_hip_reduce_film syncs first). A rank whose _begin fails on its own side (a HIP call, a failed iteration) still joins the collective - zero
 * snapshot, failed flag - and every rank gets an error from _end. The one case in which the other ranks are left to the RCCL timeout: a rank with
 * nothing to join with (no scene uploaded, or the film-sized buffers could not be ALLOCATED; they normally exist since etx_hip_comm_init /
 * etx_hip_upload_scene / etx_hip_begin). */
int etx_hip_reduce_film_begin(etx_hip_context* context);
int etx_hip_reduce_film_end(etx_hip_context* context, int wait);
int etx_hip_reduce_film(etx_hip_context* context);

typedef struct etx_hip_reduce_info_t {
  uint64_t reduces;            /* finished reduces since etx_hip_comm_init */
  uint64_t payload_bytes;      /* film bytes one reduce sums per rank (layers of the armed integrator x pixels x 16) */
  uint64_t global_iterations;  /* iterations of all ranks in the newest finished reduce (host-side count at the time of each rank's _begin) */
  double last_device_ms;       /* device time of the newest finished reduce: snapshot kernel + collectives (HIP events on the communication stream) */
  double total_device_ms;      /* ... summed over all finished reduces (reduces collected together are priced at the newest one's time) */
  uint32_t pending;            /* reduces begun and not yet collected by etx_hip_reduce_film_end */
  uint32_t layer_mask;         /* film layers of the newest reduce (bit 0 camera, 1 light, 2 normal, 3 albedo) */
} etx_hip_reduce_info_t;
int etx_hip_reduce_info(etx_hip_context* context, etx_hip_reduce_info_t* out_info, size_t info_size);

/* Two small collectives over the context's communicator for a host that has no process-group library of its own (bench.py: barrier and "the
 * slowest rank's time" without loading a second ROCm runtime into the process). Every rank calls them in the same order; they are ordered
 * behind the film reduces on the communication stream and return when the result is in `values`. op: 0 sum, 1 max, 2 min; count <= 16.
 * A context without a communicator (one rank) returns at once with `values` unchanged. */
#define ETX_HIP_REDUCE_SUM 0
#define ETX_HIP_REDUCE_MAX 1
#define ETX_HIP_REDUCE_MIN 2
int etx_hip_comm_all_reduce_f64(etx_hip_context* context, double* values, uint32_t count, int op);
int etx_hip_comm_barrier(etx_hip_context* context);

/* The HIP runtime this process runs the library on: out_versions = {hipRuntimeGetVersion() of the runtime mapped into the process, HIP_VERSION
 * the library was compiled against, RCCL version of the mapped librccl, RCCL version compiled against}. etx_hip_create refuses (ETX_HIP_ERROR_HIP)
 * a runtime whose major.minor is OLDER than the one the library was built for - what happens when a package that bundles its own ROCm runtime
 * under the same sonames (a PyTorch wheel) is imported before libetx_hip.so is loaded; load the library first, or set ETX_HIP_ALLOW_OLDER_RUNTIME=1
 * to run anyway. No context needed. */
int etx_hip_runtime_info(int out_versions[4]);

/* ------------------------------------------------------------------------------------------------------------ */
/* kernel-level entry points (used by tests/ and bench.py; the host integrator does not need them) */

/* Closest-hit query over the uploaded scene = Raytracing::trace semantics (rt.cxx:428-466) without the Intersection
 * expansion: rays = n * {ox,oy,oz,tmin,dx,dy,dz,tmax} (host), hits = n * {u, v, t, triangle_index as u32 bits},
 * triangle_index 0xffffffff when nothing was hit. Runs the production traversal kernel. */
int etx_hip_trace_rays(etx_hip_context* context, const float* rays_8f, uint64_t count, float* hits_4f);

/* Same kernel on device-resident buffers (device pointers), `repeat` launches back to back; returns the average
 * kernel time in ms measured with HIP events on the launch stream. Used by bench.py for the traversal roofline. */
int etx_hip_trace_rays_device(etx_hip_context* context, const void* d_rays_o_tmin, const void* d_rays_d_tmax, uint64_t count, void* d_hits, uint32_t repeat, double* out_avg_ms);

/* The same measurement for a host without device buffers of its own: rays as for etx_hip_trace_rays (host), uploaded once, one untimed launch,
 * then `repeat` timed launches over the device-resident queue; hits_4f (nullable) receives the hits of the last launch. */
int etx_hip_trace_rays_timed(etx_hip_context* context, const float* rays_8f, uint64_t count, uint32_t repeat, double* out_avg_ms, float* hits_4f);

/* Runs a device-side known-answer kernel: `which` selects the function, in/out are host float arrays.
 *   0: Sampler        in: n*{a,b} as u32 bits           out: n*{seed bits, next(), next(), next()}   (sampler.hxx:54-77)
 *   1: offset_ray     in: n*{p.xyz, n.xyz}              out: n*{xyz}                                 (math.hxx:925-943)
 *   2: orthonormal_basis in: n*{n.xyz}                  out: n*{u.xyz, v.xyz}                        (math.hxx:736-746)
 *   3: sample_cosine_distribution(rnd, n, 1) in: n*{rx,ry,n.xyz} out: n*{xyz}                        (math.hxx:748-762)
 *   4: grid cell_index in: n*{x,y,z as i32 bits, mask}  out: n*{index as u32 bits}                   (vcm_shared.hxx:820-822)
 *   5: sample_disk    in: n*{rx,ry}                     out: n*{x,y}                                 (math.hxx:773-790)
 *   6 + 16*set: sample_blue_noise from the uploaded table of that set: in: n*{px,py,sample as u32 bits}
 *                     out: n*{dimensions 0..5}                                                     (path_tracing.cxx:173-178)
 */
int etx_hip_kat(etx_hip_context* context, int which, const float* in, uint64_t count, float* out);

/* Self test of the traversal stack: the kernels keep 32 entries per lane in LDS and spill the rest (trees of more than ~40 000
 * triangles can need more by their worst-case bound; accepted up to 512, the spill area is sized by the uploaded tree's own bound) to global memory.
 * Every lane of a traversal-sized grid pushes `depth` (1..512)
 * values, pops half, pushes again, pops everything and compares; *out_errors = mismatches (0 expected). A real ray stays below 20
 * entries, so this is what exercises the spill. */
int etx_hip_selftest_stack(etx_hip_context* context, uint32_t depth, uint32_t* out_errors);

/* Host-only (no GPU, no context): builds the traversal BVH for `scene` exactly as etx_hip_upload_scene does and checks
 * its invariants (every triangle referenced once, child boxes enclose their triangles, leaf size <= 8, depth within the
 * device stack). out_info = {BVH4 node count, triangle count, depth | stack entries needed << 16, bytes}. Returns 0 or ETX_HIP_ERROR_INVALID_ARGUMENT. */
int etx_hip_host_check_bvh(const etx_abi_scene* scene, uint32_t out_info[4]);

/* Host-only (no GPU, no context): walks the same BVH4 in the same order as the traversal kernels for `count` rays
 * {ox,oy,oz,tmin,dx,dy,dz,tmax} and returns the work they do: out[0] node visits, out[1] triangle tests, out[2] rays that hit,
 * out[3] deepest use of the traversal stack. */
int etx_hip_host_bvh_stats(const etx_abi_scene* scene, const float* rays_8f, uint64_t count, uint64_t out[4]);

/* The same two for either builder (ETX_HIP_BVH_*). ETX_HIP_BVH_DEVICE_LBVH runs the DEVICE build on the host, element by element
 * through the functions the kernels call (dev_lbvh.h): the tree the device will build can be checked without a GPU. hits_2f
 * (nullable): per ray {t, triangle index as u32 bits (0xffffffff: miss)} of the walk. */
int etx_hip_host_check_bvh_builder(const etx_abi_scene* scene, int builder, uint32_t out_info[4]);
int etx_hip_host_bvh_stats_builder(const etx_abi_scene* scene, int builder, const float* rays_8f, uint64_t count, uint64_t out[4], float* hits_2f);

#ifdef __cplusplus
}
#endif

#endif /* ETX_HIP_H */
