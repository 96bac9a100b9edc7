/*
 * etx_scene_abi.h - byte layout of the scene data the HIP backend borrows from the etx-tracer host.
 *
 * The backend's input is the host's own `etx::Scene` / `etx::Camera` ("Inputs & ownership").
 * To keep libetx_hip.so free of any reference header, the layouts are restated here as plain C structs; every
 * struct cites the reference definition it mirrors. oracle/ref/abi_check.cxx includes BOTH this file and the
 * reference headers and static_asserts size and field offsets, so a drift in the reference breaks the oracle build.
 *
 * All cross references are u32 indices, ETX_ABI_INVALID (= etx::kInvalidIndex) means "none".
 * All pointers are HOST pointers; etx_hip_upload_scene() deep-copies what it needs.
 */
#ifndef ETX_SCENE_ABI_H
#define ETX_SCENE_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ETX_ABI_INVALID 0xffffffffu

typedef struct etx_abi_float2 { float x, y; } etx_abi_float2;
typedef struct etx_abi_float3 { float x, y, z; } etx_abi_float3;
typedef struct etx_abi_float4 { float x, y, z, w; } etx_abi_float4;
typedef struct etx_abi_uint2 { uint32_t x, y; } etx_abi_uint2;
typedef struct etx_abi_uint3 { uint32_t x, y, z; } etx_abi_uint3;

/* etx::ArrayView<T>  sources/etx/render/shared/base.hxx:52-89  {T* a; u64 count}, 16-byte aligned */
typedef struct etx_abi_array {
  const void* a;
  uint64_t count;
} __attribute__((aligned(16))) etx_abi_array;

/* etx::Vertex  sources/etx/render/shared/math.hxx:599-605 */
typedef struct etx_abi_vertex {
  etx_abi_float3 pos, nrm, tan, btn;
  etx_abi_float2 tex;
} etx_abi_vertex; /* 56 */

/* etx::Triangle  math.hxx:607-612 */
typedef struct etx_abi_triangle {
  uint32_t i[3];
  uint32_t material_index;
  etx_abi_float3 geo_n;
  float pad;
} __attribute__((aligned(16))) etx_abi_triangle; /* 32 */

/* etx::SpectralImage / SampledImage / RefractiveIndex / Thinfilm / SubsurfaceMaterial
 * sources/etx/render/shared/material.hxx:8-52, spectrum.hxx:547-551 */
typedef struct etx_abi_spectral_image { uint32_t spectrum_index, image_index; } etx_abi_spectral_image;
typedef struct etx_abi_sampled_image { etx_abi_float4 value; uint32_t image_index, channel; } etx_abi_sampled_image;
typedef struct etx_abi_refractive_index { uint32_t cls, eta_index, k_index; } etx_abi_refractive_index;
typedef struct etx_abi_thinfilm {
  etx_abi_refractive_index ior;
  uint32_t thickness_image;
  float min_thickness, max_thickness, pad;
} etx_abi_thinfilm; /* 28 */
typedef struct etx_abi_subsurface { uint32_t spectrum_index, image_index, cls, path; } etx_abi_subsurface;

/* etx::Material::Class  material.hxx:54-69 */
enum {
  ETX_MAT_DIFFUSE = 0, ETX_MAT_TRANSLUCENT, ETX_MAT_PLASTIC, ETX_MAT_CONDUCTOR, ETX_MAT_DIELECTRIC, ETX_MAT_THINFILM,
  ETX_MAT_MIRROR, ETX_MAT_BOUNDARY, ETX_MAT_VELVET, ETX_MAT_PRINCIPLED, ETX_MAT_VOID, ETX_MAT_COUNT
};

/* etx::Material  material.hxx:53-97 */
typedef struct etx_abi_material {
  etx_abi_spectral_image reflectance, scattering, emission;
  etx_abi_sampled_image roughness, metalness, transmission;
  etx_abi_subsurface subsurface;
  etx_abi_thinfilm thinfilm;
  etx_abi_refractive_index ext_ior, int_ior;
  uint32_t cls, int_medium, ext_medium, normal_image_index, diffuse_variation, two_sided;
  float normal_scale, opacity, emission_collimation;
} etx_abi_material; /* 200 */

/* etx::EmitterProfile::Class  sources/etx/render/shared/emitter.hxx:8-14 */
enum { ETX_EMITTER_AREA = 0, ETX_EMITTER_ENVIRONMENT = 1, ETX_EMITTER_DIRECTIONAL = 2 };

/* etx::EmitterProfile  emitter.hxx:7-43 */
typedef struct etx_abi_emitter_profile {
  etx_abi_spectral_image emission;
  etx_abi_float3 direction;
  uint32_t cls;
  float angular_size, equivalent_disk_size, angular_size_cosine, pad0, pad1;
} __attribute__((aligned(16))) etx_abi_emitter_profile; /* 48 */

/* etx::Emitter  emitter.hxx:45-71 */
typedef struct etx_abi_emitter {
  uint32_t cls, profile, triangle_index;
  float spectrum_weight, additional_weight, triangle_area, pad0, pad1;
} etx_abi_emitter; /* 32 */

/* etx::Distribution  sources/etx/render/shared/distribution.hxx:7-35 */
typedef struct etx_abi_distribution_entry { float value, pdf, cdf; } etx_abi_distribution_entry;
typedef struct etx_abi_distribution {
  etx_abi_array values; /* etx_abi_distribution_entry */
  float total_weight;
} __attribute__((aligned(16))) etx_abi_distribution; /* 32 */

/* etx::Image  sources/etx/render/shared/image.hxx:8-53 */
enum { ETX_IMAGE_FORMAT_UNDEFINED = 0, ETX_IMAGE_FORMAT_RGBA32F = 1, ETX_IMAGE_FORMAT_RGBA8 = 2 };
enum {
  ETX_IMAGE_BUILD_SAMPLING_TABLE = 1u << 0, ETX_IMAGE_REPEAT_U = 1u << 1, ETX_IMAGE_REPEAT_V = 1u << 2,
  ETX_IMAGE_HAS_ALPHA = 1u << 4, ETX_IMAGE_UNIFORM_SAMPLING_TABLE = 1u << 5
};
typedef struct etx_abi_image {
  etx_abi_array pixels;          /* float4 (RGBA32F) or ubyte4 (RGBA8) */
  etx_abi_array x_distributions; /* etx_abi_distribution per row */
  etx_abi_distribution y_distribution;
  etx_abi_float2 fsize, offset, scale;
  etx_abi_uint2 isize;
  float normalization;
  uint32_t options, format, data_size;
} __attribute__((aligned(16))) etx_abi_image; /* 112 */

/* etx::Medium  sources/etx/render/shared/medium.hxx:8-47 */
typedef struct etx_abi_medium {
  etx_abi_array density; /* float, dimensions.x*y*z */
  etx_abi_float3 bounds_min; float bounds_pad0;
  etx_abi_float3 bounds_max; float bounds_pad1;
  uint16_t cls; /* 0 homogeneous, 1 heterogeneous */
  uint16_t enable_explicit_connections;
  uint32_t absorption_index, scattering_index;
  float phase_function_g, max_sigma;
  etx_abi_uint3 dimensions;
} __attribute__((aligned(16))) etx_abi_medium; /* 80 */

/* etx::SpectralDistribution  sources/etx/render/shared/spectrum.hxx:447-545
 * (441 (wavelength, power) pairs, count, then the private `integrated_value` float3 = RGB/XYZ used in RGB mode) */
#define ETX_ABI_SPECTRUM_MAX_ENTRIES 441
typedef struct etx_abi_spectrum {
  struct { float wavelength, power; } entries[ETX_ABI_SPECTRUM_MAX_ENTRIES];
  uint32_t entry_count;
  etx_abi_float3 integrated;
} __attribute__((aligned(16))) etx_abi_spectrum; /* 3552 */

/* etx::Camera  sources/etx/render/shared/camera.hxx:8-39 */
typedef struct etx_abi_camera {
  float view_proj[16]; /* float4x4, column major: col[c] = view_proj[4*c .. 4*c+3] */
  etx_abi_float3 position; uint32_t cls; /* 0 perspective, 1 equirectangular */
  etx_abi_float3 target; float tan_half_fov;
  etx_abi_float3 side; float aspect;
  etx_abi_float3 up; float area;
  etx_abi_float3 direction; float image_plane;
  etx_abi_uint2 film_size;
  float lens_radius, focal_distance, clip_near, clip_far;
  uint32_t lens_image, medium_index;
} __attribute__((aligned(16))) etx_abi_camera; /* 176 */

/* etx::Scene  sources/etx/render/shared/scene.hxx:16-65 */
#define ETX_ABI_MAX_ENVIRONMENT_EMITTERS 63
enum { ETX_SCENE_COMMITTED = 1u << 0, ETX_SCENE_SPECTRAL = 1u << 1 };
typedef struct etx_abi_scene {
  etx_abi_array vertices;            /* etx_abi_vertex */
  etx_abi_array triangles;           /* etx_abi_triangle */
  etx_abi_array triangle_to_emitter; /* u32 */
  etx_abi_array materials;           /* etx_abi_material */
  etx_abi_array emitter_profiles;    /* etx_abi_emitter_profile */
  etx_abi_array emitter_instances;   /* etx_abi_emitter */
  etx_abi_array images;              /* etx_abi_image */
  etx_abi_array mediums;             /* etx_abi_medium */
  etx_abi_array spectrums;           /* etx_abi_spectrum */
  etx_abi_distribution emitters_distribution;
  struct { uint32_t emitters[ETX_ABI_MAX_ENVIRONMENT_EMITTERS]; uint32_t count; } __attribute__((aligned(16))) environment_emitters;
  etx_abi_float3 bounding_sphere_center;
  float bounding_sphere_radius;
  struct { uint32_t image_index; float radius; } pixel_sampler;
  uint32_t min_path_length, max_path_length, samples, random_path_termination;
  float noise_threshold, radiance_clamp;
  uint32_t black_spectrum, white_spectrum, rayleigh_spectrum, mie_spectrum, ozone_spectrum;
  uint32_t subsurface_scatter_material, subsurface_exit_material;
  uint32_t default_dielectric_eta, default_conductor_eta, default_conductor_k;
  uint32_t flags;
} __attribute__((aligned(16))) etx_abi_scene; /* 528 */

/* etx::VCMOptions  sources/etx/rt/shared/vcm_shared.hxx:12-72 (passed by value to etx_hip_begin) */
enum {
  ETX_VCM_CONNECT_TO_CAMERA = 1u << 0, ETX_VCM_DIRECT_HIT = 1u << 1, ETX_VCM_CONNECT_TO_LIGHT = 1u << 2,
  ETX_VCM_CONNECT_VERTICES = 1u << 3, ETX_VCM_MERGE_VERTICES = 1u << 4, ETX_VCM_ENABLE_MIS = 1u << 5,
  ETX_VCM_ENABLE_MERGING = 1u << 6,
  ETX_VCM_FULL_OPTIONS = 0x7f
};
enum { ETX_VCM_KERNEL_TOPHAT = 0, ETX_VCM_KERNEL_EPANECHNIKOV = 1 };
typedef struct etx_abi_vcm_options {
  uint32_t options, radius_decay, kernel;
  float initial_radius;
  uint8_t blue_noise;
  /* Not a member of etx::VCMOptions: it lives in that struct's tail padding (offset 17 of 32), so a VCMOptions copied over this
   * struct leaves it where the binding sets it (integration/etx_hip_integrators.hxx, option key "hip-reference_seeding").
   * 0 (default): the camera path of pixel i draws from a stream of its own. 1: it starts from the SAME sampler state as light path i,
   * as the reference does (vcm_shared.hxx:312,357) - the device then reproduces the unmodified reference's estimator wherever the
   * reference's own film is defined (its candidate draws pinned: DESIGN.md 4, tests/test_gpu_options.py); the vertex connections of a
   * pixel are correlated with its light path in that mode, which is the reference's behaviour, not a defect of the option. */
  uint8_t reference_seeding;
} __attribute__((aligned(16))) etx_abi_vcm_options; /* 32 */

/* etx::PTOptions  sources/etx/rt/shared/path_tracing_shared.hxx:8-14 */
typedef struct etx_abi_pt_options {
  uint32_t path_per_iteration;
  uint8_t nee, direct, mis, blue_noise;
} __attribute__((aligned(16))) etx_abi_pt_options; /* 16 */

/* CPUBidirectionalImpl's options  sources/etx/rt/integrators/bidirectional.cxx:323-340,1443-1466 ("bdpt-mode", "bdpt-conn_*",
 * "bdpt-blue_noise"). mode: CPUBidirectionalImpl::Mode */
enum { ETX_BDPT_MODE_PATH_TRACING = 0, ETX_BDPT_MODE_LIGHT_TRACING = 1, ETX_BDPT_MODE_FAST = 2, ETX_BDPT_MODE_FULL = 3 };
typedef struct etx_abi_bdpt_options {
  uint32_t mode;
  uint8_t direct_hit, connect_to_camera, connect_to_light, connect_vertices, mis, blue_noise;
  uint8_t reference_seeding; /* as in etx_abi_vcm_options: camera path i keeps the seed of emitter path i (bidirectional.cxx:377-378) */
} __attribute__((aligned(16))) etx_abi_bdpt_options; /* 16 */

#ifdef __cplusplus
}
#endif

#endif /* ETX_SCENE_ABI_H */
