// pipeline.h - HBM-resident wavefront state shared by the host (host_api.cpp) and the kernels.
//
// Data layout (re-derived for gfx950):
//  * Path state is SoA and is re-compacted every bounce: the shade kernel reads slot i of the "in" set and writes the
//    survivors densely into the "out" set (wave ballot + one atomic per wave), so every kernel streams contiguous
//    16-byte lanes (1 KiB per wave instruction) and the ray / hit queues are index-aligned with the state
//    (no indirection, no per-ray path index).
//  * Ray queue : ray_o_tmin[i] = {o.xyz, tmin}, ray_d_tmax[i] = {d.xyz, tmax}      32 B / ray
//  * Hit queue : hit[i] = {u, v, t, triangle index bits}                             16 B / ray
//  * Light vertices: write-once SoA pool + per-path singly linked list (bounce order != path order).
//  * Photon grid: counting sort by hash cell into a second SoA (reference: vcm_shared.cxx:49-152).
//  * Film: float4 sums per pixel (camera, light), atomically accumulated; the host divides by the iteration count,
//    which equals the reference's running-mean lerp (film.cxx:200-206, 332-343) and makes the multi-GPU reduce a sum.
#pragma once

#include "dev_scene.h"

namespace etxd {

struct PathSet {
  float4* ray_o_tmin;
  float4* ray_d_tmax;
  float4* thr_eta;   // throughput rgb, eta
  float4* mis;       // d_vcm, d_vc, d_vm, path_distance
  uint4* meta;       // sampler seed, total_path_depth, medium index, flags
  uint32_t* path_id; // global path index (light pass) / pixel index (camera pass)
  float* wavelength; // spectral mode: the path's wavelength (VCMPathState::spect)
  // bidirectional path tracing (dev_bdpt.h): summary of the previous path vertex
  float4* prev_pos;  // position, vertex flags bits
  float4* prev_nrm;  // shading normal, light path: pool slot of that vertex / camera path: its triangle
};

enum : uint32_t {  // VCMPathState flags, vcm_shared.hxx:92-98
  kPathDeltaEmitter = 1u << 0,
  kPathLocalEmitter = 1u << 3,
};

struct LightVertexPool {   // VCMLightVertex (vcm_shared.hxx:154-197): one 96-byte record per vertex (write once; a
                           // connection reads a whole vertex, so the record is interleaved: 1.5 cache lines, not 6)
  float4* rec;             // kLvStride float4 per vertex
  uint32_t capacity;
  static constexpr uint32_t kLvStride = 6;
  ETX_HD float4& pos_dvcm(uint32_t i) const { return rec[i * kLvStride + 0]; }    // pos.xyz, d_vcm
  ETX_HD float4& wi_dvc(uint32_t i) const { return rec[i * kLvStride + 1]; }      // w_i.xyz, d_vc
  ETX_HD float4& thr_dvm(uint32_t i) const { return rec[i * kLvStride + 2]; }     // throughput rgb, d_vm
  ETX_HD float4& nrm_tri(uint32_t i) const { return rec[i * kLvStride + 3]; }     // nrm.xyz, triangle index bits (kInvalid = medium vertex)
  ETX_HD float4& bc_len_med(uint32_t i) const { return rec[i * kLvStride + 4]; }  // bc.u, bc.v, (index_in_path << 16 | path_length) bits, medium bits
  ETX_HD uint32_t& next(uint32_t i) const { return reinterpret_cast<uint32_t*>(rec + i * kLvStride + 5)[0]; }  // previous vertex of the same path
  ETX_HD float& wavelength(uint32_t i) const { return reinterpret_cast<float*>(rec + i * kLvStride + 5)[1]; }  // spectral mode: the light path's wavelength
};

struct PhotonGrid {        // VCMSpatialGridData (vcm_shared.hxx:805-827), photons sorted by hash cell
  uint32_t* cell_ends;     // size hash_capacity
  float4* pos_len;         // pos.xyz, path_length bits - dense: the distance filter touches nothing else
  float4* rec;             // kPhotonStride float4 per photon, one 64-byte line for the ~20 % that pass the filter
  uint32_t* block_sums;    // scan scratch
  uint32_t hash_capacity;
  static constexpr uint32_t kPhotonStride = 4;
  ETX_HD float4& nrm_dvcm(uint32_t i) const { return rec[i * kPhotonStride + 0]; }  // nrm.xyz, d_vcm
  ETX_HD float4& win_dvm(uint32_t i) const { return rec[i * kPhotonStride + 1]; }   // w_in.xyz, d_vm
  ETX_HD float4& thr(uint32_t i) const { return rec[i * kPhotonStride + 2]; }       // throughput_rgb / sampling_pdf
};

struct GridParams {        // written by k_grid_setup on the device each iteration
  f3 bbox_min;
  float cell_size;
  f3 bbox_max;
  float radius_squared;
  float inv_radius_squared;
  uint32_t hash_mask;
  uint32_t photon_count;
  uint32_t valid;
};

struct CameraVertexPool {  // connectible camera vertices of the current bounce (input of k_connect / k_merge)
  float4* hit;             // u, v, t, triangle bits  (medium event: pos.xyz, kInvalid)
  float4* wi_medium;       // ray direction (w_i), medium index bits
  float4* thr_depth;       // throughput rgb, total_path_depth bits
  float4* mis_pixel;       // d_vcm, d_vc, d_vm (already updated at the vertex), pixel index bits
  uint32_t* seed;
  float* wavelength;       // spectral mode: wavelength of the camera path
  // merge-ready copy (k_merge reads these five float4 and nothing else): no dependent triangle/vertex/material loads
  float4* pos_info;        // pos.xyz, (total_path_depth << 8) | flags   (kCvDiffuse, kCvMedium)
  float4* nrm_dvm;         // shading normal, d_vm
  float4* fthr_dvcm;       // diffuse: albedo/pi * throughput (= func * t_camera), else throughput; d_vcm
};

// kCvNoMerge / kCvNoConnect: the exit points of a Christensen-Burley vertex are connected one by one, the photon merge happens once at
// the exit point the path continues from, with another throughput (vcm_shared.hxx:1038-1071): two kinds of records
enum : uint32_t { kCvDiffuse = 1u << 0, kCvMedium = 1u << 1, kCvNoMerge = 1u << 2, kCvNoConnect = 1u << 3 };

// Endpoint connections of the general / subsurface shading groups: the connection of a light vertex to the camera
// (vcm_connect_to_camera, vcm_shared.hxx:463-535) or of a camera vertex to a light (vcm_connect_to_light, :608-671).
// Both evaluate the vertex' BSDF three times (evaluate, reverse pdf, pdf - stochastic walks for the Heitz models); inside
// the step kernels those calls kept the whole path state alive across them. The step functions of these groups write
// a 100-byte request instead and k_connect_endpoints evaluates the requests densely, one per lane, then queues the
// visibility segment like every other connection. The simple group (Lambert / delta) still connects inline.
struct EndpointQueue {
  float4* hit;        // u, v, t, triangle bits  (medium vertex: pos.xyz, kInvalid)
  float4* wi_medium;  // w_i at the vertex, medium index bits
  float4* thr_depth;  // throughput rgb (scaled by a subsurface walk, if any), total_path_depth | kCvExitMaterialBit
  float4* mis_id;     // d_vcm, d_vc, unused, path id bits (pixel index in the camera pass)
  float4* rnd_seed;   // the three fixed randoms of the connection (rnd_connection.xy, rnd_support.y), sampler seed bits
  float* wavelength;
  uint32_t capacity;
};
constexpr uint32_t kCvGeneralBsdfBit = 0x40000000u;   // in thr_depth.w (bidirectional records): the vertex' material is not of the simple shading group
constexpr uint32_t kCvExitMaterialBit = 0x80000000u;  // in thr_depth.w: the vertex is the exit point of a subsurface walk, material = scene.subsurface_exit_material

struct ShadowQueue {        // transmittance ("shadow") ray requests of the current bounce: 48 B in, film atomics out
  float4* p0_medium;       // segment start, medium index bits at the start
  float4* p1_target;       // segment end, film target bits: bit 31 = light layer, low bits = film pixel index
  float4* value;           // contribution before transmittance (rgb, already weighted for the film), w = wavelength (media)
  uint32_t capacity;
};

constexpr uint32_t kShadowTargetLight = 0x80000000u;
constexpr uint32_t kPathTableEntries = 32;         // Pipeline::path_table_entries of scenes without subsurface materials: k_expand_pairs reads a light path's first entries from the
                                                   // table (16-byte loads) and WALKS the list beyond it, one dependent gather per vertex - a wavefront waits for its longest path. Fog box
                                                   // (3.8 vertices per path, long tail): 8 entries 101.7, 16 103.0, 32 103.9 Msamples/s (profiles/round4_ab_medium_rows_and_path_table.txt)
constexpr uint32_t kPathTableEntriesWalk = 32;     // ... with: a light path that walked through an object has a vertex per scattering event
constexpr uint32_t kPathRowHeader = 2;             // words 0 / 1 of a path's table row: head, length (VCM records)
// The bidirectional integrator's rows: [0] the path's newest overflow chunk (kInvalid = none), [1] vertices stored so far, [2] how many of them are of a general BSDF class,
// then the path's first path_table_entries - 3 vertices: pool index | kPathEntryGeneralBit. Vertices beyond the row go to CHUNKS of kPathChunkWords words (Pipeline::path_chunks:
// [0] the previous chunk, then 31 entries) - k_bdpt_expand_pairs reads a path's vertices with independent 16-byte loads however long the path is (until round 5 it walked the
// vertices' `prev` links beyond the row, one dependent gather per vertex: a light path that crossed a subsurface object has a vertex per scattering event, up to 1024 per walk).
constexpr uint32_t kBdptRowHeader = 3;
constexpr uint32_t kPathChunkWords = 32;
constexpr uint32_t kPathChunkEntries = kPathChunkWords - 1u;
constexpr uint32_t kPathEntryGeneralBit = 0x80000000u;
constexpr uint32_t kMergeBucketBits = 6;                                  // per axis
constexpr uint32_t kMergeBuckets = 1u << (3u * kMergeBucketBits);         // 64^3 coarse buckets

// Device counters (u32), cleared per iteration unless noted. Atomics on one 128-byte line are serialised by the L2
// channel that owns it - measured 88 M atomics/s chip-wide (11.3 ns each) no matter how many different words of the
// line are hit, 3x that for three lines - so every hot counter lives on a line of its own (index stride 32) and the
// queue producers reserve slots once per workgroup, not once per wavefront (dev_vcm.h block_compact_slot).
enum : uint32_t {
  kCntActiveA = 0,
  kCntActiveB = 32,
  kCntLightVertices = 64,
  kCntCameraVertices = 96,   // cleared per bounce
  kCntOverflow = 128,
  kCntBboxMin = 160,         // 3 x ordered-int float min
  kCntBboxMax = 192,         // 3 x ordered-int float max
  kCntPairs = 224,           // (camera vertex, light vertex) pairs of the current bounce, cleared per bounce
  kCntShadow = 256,          // shadow requests of the current bounce, cleared per bounce
  kCntMergeVertices = 288,   // camera vertices of the current bounce that take part in the merge
  kStatRaysExtension = 320,  // u64 statistics
  kStatRaysShadow = 352,
  kStatCameraVertices = 384,
  kStatPhotonsExamined = 416,
  kStatPhotonsMerged = 448,
  kStatSplats = 480,
  kDbgBase = 512,            // u64 debug counters (ETX_HIP_DEBUG_COUNTERS builds)
  kCntGroupGeneral = 544,    // paths of the current bounce whose hit material is shaded by the general kernel, cleared per bounce
  kCntGroupSubsurface = 576, // ... by the subsurface kernel
  kCntEndpoints = 608,       // endpoint connection requests of the current bounce, cleared per bounce
  kCntNonFinite = 640,
  kCntLightBounceBegin = 800, // BDPT: light vertex count when the current bounce began (k_bdpt_connect_camera covers [begin, count))
  kCntPairsGeneral = 802,     // BDPT: pairs with a vertex of a general BSDF class, listed from the BACK of the pair buffer (kCntPairs counts the others, from the front); cleared per bounce
  kCntPathChunks = 803,       // BDPT: chunks of the light paths' overflow index lists handed out in this iteration (Pipeline::path_chunks)
  kCntDynMedium = 801,        // BDPT: per-walk medium rows appended in this iteration (DScene::sss_dynamic_media; one atomic per wavefront, shares a line with a word written once per bounce)
  kStatRaysLight = 672,      // u64 statistics: closest-hit rays of the light pass / camera pass, pair connections, endpoint connections
  kStatRaysCamera = 704,
  kStatPairs = 736,
  kStatEndpoints = 768,       // film contributions dropped because they were not finite (never cleared within a run: reported by etx_hip_stats)
  kStatActivePixels = 832,  // u64: pixels sampled (path tracing with adaptive sampling: Film::active_pixel)
  kCntWalk = 864,            // BDPT: entries of walk queue 0 / 1 (+ 32): paths inside a subsurface object. The round whose input path set is s takes
                             // its walks from queue s (new ones of this bounce + the unfinished ones of the previous round) and leaves the walks
                             // that are still unfinished after kWalkBudget events in queue s ^ 1
  kCntWalkFetch = 928,       // ... and how many entries the walk kernel's wavefronts have taken in this round
  kCntWalkExit = 960,        // ... and the walks that have reached their object's surface (exit queue, k_bdpt_walk_exit), cleared per bounce
  kStatCrossings = 992,      // u64: closest-hit queries the traversal kernel ran BEYOND a medium boundary it crossed itself (kernels_trace.hip kCross); part of kStatRaysExtension
  kCounterCount = 1024,
};

// Per-workgroup statistics (u64): workgroup b of any launch adds to row b without atomics (launches on one stream do
// not overlap); k_stats_finalize folds the rows into the counters above once per iteration.
enum : uint32_t { kBlockStatExamined = 0, kBlockStatMerged = 1, kBlockStatSplats = 2, kBlockStatCrossings = 3, kBlockStatCount = 4 };
constexpr uint32_t kBlockStatRows = 2048 + 8;

enum : uint32_t {
  kOverflowLightVertices = 1u << 0,
  kOverflowStack = 1u << 1,
  kOverflowPairs = 1u << 2,
  kOverflowShadow = 1u << 3,
  kOverflowCameraVertices = 1u << 4,
  kOverflowEndpoints = 1u << 5,
};

struct VcmParams {  // VCMOptions + VCMIteration (vcm_shared.hxx:12-89), per iteration, by value
  uint32_t options;
  uint32_t kernel;
  uint32_t iteration;
  uint32_t path_count;   // paths per sub pass: W * H, or this context's share of the pixels (pixel_first / pixel_stride)
  float current_radius;
  float vm_weight;
  float vc_weight;
  float vm_normalization;
  uint32_t film_w, film_h;
  const uint2* bluenoise;  // [128*128][256] x 8 bytes (etx_hip_upload_bluenoise), nullptr = options.blue_noise off
  // pixel-interleaved sharding of the path tracer and the bidirectional integrator (etx_hip_begin_ex): path k of this context belongs to
  // pixel pixel_first + k * pixel_stride; VCM always runs (0, 1) - a photon map needs the light paths of every pixel
  uint32_t pixel_first, pixel_stride;
};

// the pixel (= path id: sampler seed, film position, per-path tables) of the k-th path a context generates
ETX_HD uint32_t path_pixel(const VcmParams& it, uint32_t k) { return it.pixel_first + k * it.pixel_stride; }

// VcmParams::options bit 31 (bidirectional integrator): this is the second attempt at an iteration whose pools overflowed - the first attempt
// was not committed (k_vcm_commit) but has already added the iteration's normal / albedo values, which go straight to the film
constexpr uint32_t kOptionRetryKeepsAovs = 1u << 31;
// VcmParams::options bit of etx_abi_vcm_options / _bdpt_options::reference_seeding: camera path i starts from the sampler state of light path i
// (vcm_shared.hxx:312,357; bidirectional.cxx:377-378) instead of a stream of its own (k_camera_generate, k_bdpt_camera_generate)
constexpr uint32_t kOptionReferenceSeeding = 1u << 30;

// PT: VcmParams::options carries PTOptions (path_tracing_shared.hxx:8-14)
enum : uint32_t { ETX_PT_DIRECT = 1u << 0, ETX_PT_NEE = 1u << 1, ETX_PT_MIS = 1u << 2 };

ETX_HD bool opt_connect_to_camera(const VcmParams& p) { return p.options & ETX_VCM_CONNECT_TO_CAMERA; }
ETX_HD bool opt_direct_hit(const VcmParams& p) { return p.options & ETX_VCM_DIRECT_HIT; }
ETX_HD bool opt_connect_to_light(const VcmParams& p) { return p.options & ETX_VCM_CONNECT_TO_LIGHT; }
ETX_HD bool opt_connect_vertices(const VcmParams& p) { return p.options & ETX_VCM_CONNECT_VERTICES; }
ETX_HD bool opt_enable_mis(const VcmParams& p) { return p.options & ETX_VCM_ENABLE_MIS; }
ETX_HD bool opt_enable_merging(const VcmParams& p) { return p.options & ETX_VCM_ENABLE_MERGING; }
ETX_HD bool opt_merge_vertices(const VcmParams& p) { return opt_enable_merging(p) && (p.options & ETX_VCM_MERGE_VERTICES); }

struct Pipeline {  // everything a kernel needs, passed by value (fits the kernarg segment)
  DScene scene;          // by value: travels in the kernel argument segment, so its fields are scalar loads and its
                         // table pointers are known global pointers (not flat) to the compiler
  PathSet paths[2];
  PathSet walk[2];       // BDPT walk queues (kCntWalk): the state of the paths that are inside a subsurface object (k_bdpt_walk_* take them from
                         // here; a path joins the "out" set again when it has left the object); null without such materials
  uint2* walk_info[2];   // ... their object: (material | events of the walk so far << 20, medium of the walk: DScene::material_sss_medium of the material, or the walk's own row)
  PathSet walk_exit;     // BDPT exit queue: walks whose free flight reached the surface of their object, with that hit
  float4* walk_exit_hits;
  float4* hits;          // hit queue, aligned with the "in" path set
  LightVertexPool lv;
  float* path_wavelength;      // spectral mode: wavelength of light path i, reused by camera path i (vcm_cpu.cxx:186)
  uint4* light_path_table;     // per path: ONE ROW of path_table_entries words: [0] the last stored vertex (kInvalid = none), [1] the vertices stored so far, then the path's
                               // first path_table_entries - 2 vertices by index in path (k_expand_pairs reads them with independent 16-byte loads instead of walking the list
                               // from the head). Head and length lived in arrays of their own until round 5: storing a vertex then dirtied three cache lines in three
                               // allocations with 4-byte writes (each a read-modify-write of its line in L2: PMC showed k_light_shade writing 1.6x and the bidirectional
                               // light shading 2.2x its algorithmic bytes) and k_expand_pairs gathered from all three per camera vertex. One row = one line.
  uint32_t* path_chunks;        // BDPT: overflow chunks of the light paths' index lists (kPathChunkWords words each), one pool per lane
  uint32_t path_chunk_capacity;
  uint32_t path_table_entries; // words per row: kPathTableEntries or kPathTableEntriesWalk (a multiple of four; two of them are the header)
  PhotonGrid grid;
  GridParams* grid_params;
  CameraVertexPool cv;
  ShadowQueue shadow;
  EndpointQueue endpoints;
  uint32_t* group_list[kShadeGroupCount - 1];  // path slots of the current bounce binned by shading group (general, subsurface); kernels_shade.inl
  uint32_t* merge_order;       // camera vertex slots of the current bounce sorted by coarse spatial bucket (k_merge_*)
  uint32_t* merge_buckets;     // kMergeBuckets + 1 counters / offsets, then 256 scan-group totals
  uint2* pairs;          // (camera vertex slot, light vertex index) of the current bounce
  uint32_t pair_capacity;
  // film: per-pixel sums over this rank's iterations, four layers in ONE allocation (camera, light, normal, albedo)
  // so that the multi-GPU film reduce is a single RCCL all-reduce
  float4* camera_sum;    // Film::CameraImage x iterations (PT kernels get the iteration image here, host_api.cpp)
  float4* light_sum;     // Film::LightImage x iterations
  float4* normal_sum;    // Film::Normals x iterations (PT)
  float4* albedo_sum;    // Film::Albedo x iterations (PT)
  // adaptive sampling (Film::estimate_noise_levels / active_pixel, film.cxx:233-330,434-459), path tracing with noise_threshold > 0
  float4* adaptive_sum;  // sum of the EVEN samples of each pixel, their count in w (StorageCameraAdaptive holds their mean)
  uint32_t* pixel_state; // bit 0 converged, bit 1 tmp (Film's internal_data); nullptr = every pixel is sampled every iteration
  uint32_t* counters;
  unsigned long long* block_stats;  // kBlockStatRows x kBlockStatCount
  uint32_t debug_flags;             // ETX_HIP_DEBUG_FLAGS: ablation switches for kernel timing experiments (0 in production)
  uint32_t capacity;     // paths per set
  uint32_t cv_capacity;  // camera vertex records per bounce: = capacity, x 8 when the scene has Christensen-Burley materials (up to 24 exit points per vertex)
};

}  // namespace etxd
