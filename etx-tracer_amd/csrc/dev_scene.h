// dev_scene.h - the scene as it lives in HBM, plus the geometry/texture accessors the kernels share.
//
// Geometry, materials, emitters stay in the reference's own POD layout (include/etx_scene_abi.h) so the host
// upload is a deep copy with pointer patching (the author's own plan, sources/etx/rt/rt.cxx:141-238). What is
// re-laid-out for the GPU: the BVH (dev_bvh.h), spectra (pre-resolved to one float4 RGB per spectrum in RGB mode),
// images (always float4 + flat CDF tables) and media.
#pragma once

#include "dev_math.h"
#include "../../include/etx_scene_abi.h"

namespace etxd {

// BVH2 node, Aila-Laine style: a node holds the boxes of BOTH children, so one 64-byte fetch decides both.
// child >= 0: inner node index; child < 0: leaf, ~child = (first_triangle << 3) | (count - 1), count <= 8.
struct __attribute__((aligned(16))) BvhNode {
  float4 lo0_hi0x;    // c0.min.xyz, c0.max.x
  float4 hi0yz_lo1xy; // c0.max.y, c0.max.z, c1.min.x, c1.min.y
  float4 lo1z_hi1;    // c1.min.z, c1.max.xyz
  int32_t child0, child1;
  uint32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be one 64-byte line");

// BVH4 node as the device traverses it: the host collapses the binned-SAH BVH2 (above: the builder's intermediate form)
// into four-wide nodes, children's boxes stored component-wise so that one node fetch (8 x 16 B) decides four children
// with four-wide slab arithmetic. Half the node visits and dependent fetches of the BVH2; nodes are numbered breadth
// first, so the first N nodes are the top of the tree - the part the traversal kernels stage in LDS (dev_bvh.h).
// child: >= 0 inner node index, < 0 leaf (~child = (first_triangle << 3) | (count - 1)), kBvhEmptyChild = unused slot.
struct __attribute__((aligned(16))) Bvh4Node {
  float4 lo_x, lo_y, lo_z, hi_x, hi_y, hi_z;
  int32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node is two 64-byte lines");
constexpr int32_t kBvhEmptyChild = 0x7fffffff;


// Triangle in traversal order: v0 + two edges, original index, filter flags.
struct __attribute__((aligned(16))) BvhTri {
  float4 v0_index;  // v0.xyz, triangle index (u32 bits)
  float4 e1_flags;  // e1.xyz, flags (u32 bits)
  float4 e2_mat;    // e2.xyz, material index (u32 bits)
};
static_assert(sizeof(BvhTri) == 48, "BvhTri");

enum : uint32_t {
  kTriVoid = 1u << 0,        // Material::Class::Void: never reported (rt.cxx:441-444)
  kTriAlphaTested = 1u << 1, // opacity < 1 or alpha texture: stochastic alpha test (scene_bsdf.hxx:128-144)
  kTriBoundary = 1u << 2,    // Material::Class::Boundary: transparent to transmittance rays (rt.cxx:503)
  kTriQuad = 1u << 3,        // flat-sweep primitive that covers a parallelogram = two triangles (FlatPrimInfo)
};

// Flat sweep (dev_bvh.h): the primitives are the scene's triangles, except that two coplanar triangles which form a
// parallelogram (same material, same filter flags, same winding - every wall of a Cornell box) are tested once as
// P = v0 + a e1 + b e2 with a, b in [0, 1]. After the sweep the hit is handed back as (triangle, u, v) of the
// reference's convention: u and v of the triangle on either side of the diagonal are affine in (a, b).
// One primitive of the sweep, pre-transformed on the host: the plane (N . X + nd = 0) and the two rows that map a point
// of the plane to the parallelogram coordinates (a = U . X + ud, b = V . X + vd). The per-ray test is 17 FMAs and one
// reciprocal instead of the 31 of Moeller-Trumbore on (v0, e1, e2).
struct __attribute__((aligned(16))) FlatPrim {
  float4 plane;   // N.xyz (= e1 x e2), nd = -N . v0
  float4 row_a;   // U.xyz = (e2 x N) / |N|^2, ud = -U . v0
  float4 row_b;   // V.xyz = (N x e1) / |N|^2, vd = -V . v0
  uint32_t flags, material;
  uint32_t medium_against, medium_along;  // Boundary primitives: the medium beyond the plane for a ray with N . d < 0 / >= 0 (host_scene.cpp)
};
static_assert(sizeof(FlatPrim) == 64, "FlatPrim");

struct __attribute__((aligned(16))) FlatPrimInfo {
  uint32_t tri_a, tri_b;  // tri_b == kInvalid: a single triangle
  float ua[3], va[3];     // a + b <= 1: u = ua[0] + ua[1] a + ua[2] b, v likewise
  float ub[3], vb[3];     // a + b >  1: barycentrics in tri_b
  uint32_t pad[2];
};
static_assert(sizeof(FlatPrimInfo) == 64, "FlatPrimInfo");

struct DImage {
  const float4* pixels;                  // isize.x * isize.y, RGBA8 sources are expanded at upload
  const etx_abi_distribution_entry* x_entries;  // isize.y rows, x_stride entries each (sampling tables only)
  const etx_abi_distribution_entry* y_entries;  // y_count entries
  uint32_t x_stride, y_count;
  f2 fsize, offset, scale;
  uint32_t isize_x, isize_y;
  float normalization;
  uint32_t options;
};

// The first three 16-byte rows hold what a homogeneous RGB query needs - one float4 load each instead of a dword per field (a lane's loads
// cost per instruction, not per byte: DESIGN.md 3). Filled by pack_rows() before the upload (host_scene.cpp).
struct alignas(16) DMedium {
  f3 absorption;  // RGB-resolved (RGB mode)
  float g;
  f3 scattering;
  uint32_t cls_explicit;  // cls | explicit_connections << 16
  f3 extinction;          // absorption + scattering
  uint32_t cls_copy;
  const float* density;
  f3 bounds_min, bounds_max;
  uint32_t absorption_index, scattering_index;  // spectra (spectral mode), kInvalid = zero
  uint32_t cls, explicit_connections;
  float max_sigma;
  uint32_t dim_x, dim_y, dim_z;
  // spectral scenes, walk medium DERIVED from a subsurface material (bidirectional integrator, host_scene.cpp): the spectra of the material's
  // colour and scattering distances; the coefficients at a wavelength are subsurface::remap_channel of their values there
  uint32_t derived_color, derived_distances;  // spectrum indices, kInvalid = not a derived medium
  void pack_rows() {
    cls_explicit = cls | (explicit_connections ? 0x10000u : 0u);
    extinction = {absorption.x + scattering.x, absorption.y + scattering.y, absorption.z + scattering.z};
    cls_copy = cls;
  }
};
static_assert(sizeof(DMedium) % 16 == 0, "rows of the medium table are read as float4");

// rows 0-1 of a medium: the free-flight sampler and the phase function (RGB mode, homogeneous)
struct MediumRows {
  f3 absorption, scattering;
  float g;
  uint32_t cls;
  bool explicit_connections;
};
ETX_DEV MediumRows load_medium_rows(const DMedium& m) {
  const float4* rows = reinterpret_cast<const float4*>(&m);
  const float4 a = rows[0], b = rows[1];
  const uint32_t bits = __float_as_uint(b.w);
  return {{a.x, a.y, a.z}, {b.x, b.y, b.z}, a.w, bits & 0xffffu, (bits & 0x10000u) != 0u};
}

struct DCamera {
  float view_proj[16];
  f3 position, side, up, direction;
  float tan_half_fov, aspect, area, image_plane;
  uint32_t film_w, film_h, cls;
  float lens_radius, focal_distance, clip_near, clip_far;
  uint32_t lens_image, medium_index;
};

// Shading groups: after the closest-hit query a path is shaded by the kernel of its hit material's group, so one rough
// gem does not move every Lambert wall hit of the scene onto the general-material kernel (per path, not per scene).
enum : uint32_t {
  kShadeGroupSimple = 0,      // Lambert Diffuse, Translucent, Mirror, Boundary, Void, roughness-0 Conductor without thin film; also misses and medium events
  kShadeGroupGeneral = 1,     // every other class (Heitz walks, thin film, Plastic, Velvet, Principled, rough-diffuse variations)
  kShadeGroupSubsurface = 2,  // any class with a subsurface layer: the random walk runs inside the shade kernel
  kShadeGroupCount = 3,
};

constexpr uint32_t kSssMediumDynamic = 0xfffffffeu;          // DScene::material_sss_medium: the walk medium is derived per entry point (textured colour / distances)
constexpr uint32_t kMediumStoredCoefficients = 0xfffffffeu;  // DMedium::derived_color of such a row: absorption / scattering hold the coefficients as they are, also in spectral mode

struct DScene {
  const etx_abi_vertex* vertices;
  const etx_abi_triangle* triangles;
  const uint32_t* triangle_to_emitter;
  const float4* tri_shade;  // per triangle: kTriShadeStride 16-byte rows of everything a shading point reads of its triangle (k_build_tri_shade)
  const etx_abi_material* materials;
  const etx_abi_emitter_profile* emitter_profiles;
  const etx_abi_emitter* emitters;
  const etx_abi_distribution_entry* emitter_dist;
  const float4* spectrum_rgb;  // RGB mode: SpectralDistribution::integrated_value per spectrum index
  // spectral mode (scene.spectral()): every spectrum as (wavelength, power) pairs, the CIE observer for the film
  const float2* spectrum_entries;
  const uint2* spectrum_ranges;  // per spectrum: first entry, entry count
  const float4* cie_xyz;         // spectrum::spectral_xyz(i), i = wavelength - cie_first (etx_hip_upload_cie_table)
  const float4* rgb_response;    // rows of rgb_response's table, i = wavelength - rgb_response_first (etx_hip_upload_rgb_response)
  const DImage* images;
  const DMedium* mediums;
  const Bvh4Node* bvh_nodes;
  const BvhTri* bvh_tris;
  const FlatPrim* flat_prims;      // bvh_flat scenes: pre-transformed primitives of the sweep
  const FlatPrimInfo* flat_info;   // per primitive
  const DScene* self;              // device address of the device-resident copy of this struct (out-of-line BSDF calls, dev_bsdf_ool.h)
  const uint32_t* material_variants;  // per material: first of its three PrincipledBSDF variants (appended to `materials`), kInvalid otherwise
  const uint8_t* material_group;   // per material: shading group of a path that hits it (kShadeGroup*, kernels_shade.inl)
  const uint8_t* material_general_bsdf;  // per material: != 0 = its BSDF calls need the out-of-line library (dev_bsdf_ool.h); the bidirectional kernels of mixed scenes split their items by it
  const uint32_t* material_sss_medium;  // per material: the medium its subsurface walk runs through (interior medium, or a derived entry appended to `mediums`; host_scene.cpp)
  uint32_t vertex_count, triangle_count, material_count, emitter_count, emitter_dist_count, spectrum_count, image_count, medium_count;
  uint32_t bvh_node_count, bvh_tri_count, flat_prim_count;
  uint32_t boundary_materials;  // materials of Class::Boundary in the table: 0 = a transmittance query is a pure occlusion test (dev_bvh.h bvh_occluded)
  uint32_t normal_mapped_materials;  // materials with a normal map: 0 = a shading point is the interpolated vertex (make_intersection reads no material for it)
  uint32_t textured_materials;  // materials that reference any image (normal map included): 0 = no BSDF reads a texture coordinate
  uint32_t heterogeneous_mediums;  // media with a density grid: 0 = every medium_transmittance is one exp (selects the lean shadow kernel together with boundary_materials)
  int32_t bvh_root;  // child encoding (a single leaf scene has a negative root)
  uint32_t bvh_depth; // levels of inner BVH4 nodes
  uint32_t bvh_stack_need;  // stack entries the traversal can need (host bound over the tree): selects the kernel variant
  int32_t* stack_spill;     // trees that need more than the LDS stack: [level - kStackDepth][lane], PER DEVICE LANE (set in the lane's Pipeline::scene copy, host_api.cpp)
  uint32_t stack_spill_lanes;
  uint32_t bvh_flat; // != 0: so few triangles that the wave-uniform linear sweep beats the tree (dev_bvh.h)
  float emitter_dist_total;
  uint32_t env_emitters[ETX_ABI_MAX_ENVIRONMENT_EMITTERS];
  uint32_t env_count;
  f3 bounds_center;
  float bounds_radius;
  uint32_t min_path_length, max_path_length, samples, random_path_termination;
  float radiance_clamp;
  uint32_t flags;
  uint32_t pixel_sampler_image;
  float pixel_sampler_radius;
  uint32_t subsurface_exit_material, subsurface_scatter_material;
  uint32_t default_dielectric_eta, default_conductor_eta, default_conductor_k;  // spectrum indices (PrincipledBSDF)
  uint32_t spectral;    // Scene::spectral(): one wavelength per path, SpectralResponse = one float (kept replicated in xyz here)
  uint32_t cie_count;
  float cie_first, cie_y_scale;  // spectrum::kShortestWavelength, 1 / kYIntegral
  uint32_t rgb_response_count;
  float rgb_response_first;
  // Subsurface materials WITHOUT an interior medium whose colour or scattering distances are TEXTURED, under the bidirectional integrator: the medium
  // of a walk is then a property of the point where the path entered (subsurface_step, bidirectional.cxx:757-771: apply_image at intersection.tex).
  // Such a walk appends a row of its own to the medium table when it starts - rows [dyn_medium_first, + dyn_medium_capacity) of `mediums`, which in
  // that case is a per-lane copy of the table (host_api.cpp allocate_pools) - and names it by index like every other medium: the walk's free flights,
  // the medium vertices it stores and the connections from them need nothing else. material_sss_medium holds kSssMediumDynamic for these materials.
  uint32_t sss_dynamic_media;   // != 0: the scene holds such a material
  uint32_t dyn_medium_first, dyn_medium_capacity;  // per lane (set in the lane's Pipeline::scene copy)
  uint32_t* lane_counters;      // per lane: Pipeline::counters (the row counter kCntDynMedium, the overflow word)
  DCamera camera;
};

// ---------------------------------------------------------------------------------------------------------------
// small loaders (the ABI structs are 4-byte aligned AoS; these compile to dword loads served by L1/L2)
ETX_DEV f3 ld3(const etx_abi_float3& v) {
  return {v.x, v.y, v.z};
}
ETX_DEV f3 spectrum_rgb(const DScene& s, uint32_t index) {
  float4 v = s.spectrum_rgb[index];
  return {v.x, v.y, v.z};
}

// scene.spectrums[index](spect), spectrum.hxx:462-504. RGB mode: the integrated value. Spectral mode: the power at the
// path's wavelength (binary search + lerp), replicated into the three components - every SpectralResponse operation of
// the hot path (products, luminance = monochromatic(), maxima, the per-channel medium sampling) then yields the
// scalar result of the reference's spectral branch, and RGB and spectral paths share one instruction stream.
ETX_DEV f3 spectrum_eval(const DScene& s, uint32_t index, float wavelength) {
  if (s.spectral == 0u)
    return spectrum_rgb(s, index);
  const uint2 range = s.spectrum_ranges[index];
  const uint32_t count = range.y;
  if (count == 0u)
    return f3{0.0f, 0.0f, 0.0f};
  const float2* entries = s.spectrum_entries + range.x;
  uint32_t b = 0, e = count;
  do {
    uint32_t m = b + (e - b) / 2;
    if (entries[m].x > wavelength)
      e = m;
    else
      b = m;
  } while ((e - b) > 1);
  const uint32_t i = b;
  const float2 ei = entries[i];
  if ((i == 0u) && (wavelength < ei.x))
    return f3{0.0f, 0.0f, 0.0f};
  if ((i + 1u == count) && (wavelength > ei.x))
    return f3{0.0f, 0.0f, 0.0f};
  const uint32_t j = min(i + 1u, count - 1u);
  const float2 ej = entries[j];
  const float t = (i == j) ? 0.0f : (wavelength - ei.x) / (ej.x - ei.x);
  const float power = ei.y + (ej.y - ei.y) * t;
  return f3{power, power, power};
}

// What a scalar contribution at `wavelength` adds to the RGB film: (value / sampling_pdf).to_rgb(), spectrum.hxx:221-223,
// 268-289 (CIE observer, xyz -> rgb). RGB mode: 1.
ETX_DEV f3 spectral_film_weight(const DScene& s, float wavelength) {
  if (s.spectral == 0u)
    return f3{1.0f, 1.0f, 1.0f};
  const float last = s.cie_first + float(s.cie_count - 1u);
  if ((wavelength < s.cie_first) || (wavelength > last))
    return f3{0.0f, 0.0f, 0.0f};
  const float w = floorf(wavelength);
  const float dw = wavelength - w;
  const uint32_t i = uint32_t(w - s.cie_first);
  const uint32_t j = min(i + 1u, s.cie_count - 1u);
  const float4 a = s.cie_xyz[i], b = s.cie_xyz[j];
  const f3 xyz = f3{a.x + (b.x - a.x) * dw, a.y + (b.y - a.y) * dw, a.z + (b.z - a.z) * dw} * s.cie_y_scale;
  const f3 rgb = {3.24045420f * xyz.x - 1.5371385f * xyz.y - 0.4985314f * xyz.z, -0.9692660f * xyz.x + 1.8760108f * xyz.y + 0.0415560f * xyz.z,
    0.05564340f * xyz.x - 0.2040259f * xyz.y + 1.0572252f * xyz.z};
  const float c = coshf(0.0072f * (wavelength - 538.0f));
  const float sampling_pdf = 0.0039398042f / (c * c);
  return rgb / sampling_pdf;
}

// absorption / scattering coefficients of a medium at the path's wavelength (RGB mode: the resolved RGB values)
// subsurface::remap_channel, scene_bssrdf_subsurface.hxx:17-44 (van de Hulst style albedo inversion)
ETX_DEV void sss_remap_channel(float color, float scattering_distance, float& albedo, float& extinction, float& scattering) {
  const float a = 1.826052378200f, b = 4.985111943850f + 0.12735595943800f, c = 1.096861024240f;
  const float d = 0.496310210422f, e = 4.231902997010f + 0.00310603949088f, f = 2.406029994080f;
  const float kMinScattering = 1.0f / 1024.0f;
  color = fmaxf(0.0f, color);
  const float blend = powf(color, 0.25f);
  albedo = (1.0f - blend) * a * powf(atanf(b * color), c) + blend * d * powf(atanf(e * color), f);
  albedo = fminf(fmaxf(albedo, 0.0f), 1.0f - kEpsilon);
  extinction = 1.0f / fmaxf(scattering_distance, kMinScattering);
  scattering = extinction * albedo;
}

ETX_DEV void medium_coefficients(const DScene& s, const DMedium& m, float wavelength, f3& absorption, f3& scattering) {
  if ((s.spectral == 0u) || (m.derived_color == kMediumStoredCoefficients)) {  // ... a per-walk row (DScene::sss_dynamic_media) holds its coefficients at the walk's own wavelength
    absorption = m.absorption;
    scattering = m.scattering;
    return;
  }
  if (m.derived_color != kInvalid) {  // subsurface_to_medium_instance / subsurface_step at this wavelength (bidirectional.cxx:729-771)
    float albedo, extinction, scatter;
    sss_remap_channel(spectrum_eval(s, m.derived_color, wavelength).x, spectrum_eval(s, m.derived_distances, wavelength).x, albedo, extinction, scatter);
    scattering = f3{scatter, scatter, scatter};
    absorption = f3{extinction - scatter, extinction - scatter, extinction - scatter};
    return;
  }
  absorption = (m.absorption_index == kInvalid) ? f3{0.0f, 0.0f, 0.0f} : spectrum_eval(s, m.absorption_index, wavelength);
  scattering = (m.scattering_index == kInvalid) ? f3{0.0f, 0.0f, 0.0f} : spectrum_eval(s, m.scattering_index, wavelength);
}

// ---- heterogeneous media: scene_medium.hxx:7-97 (bounds in the medium's local [0,1]^3 frame, trilinear density) ----
ETX_DEV bool medium_bounds(const f3& in_pos, const f3& in_dir, float max_t, float& t_min, float& t_max) {  // :12-40
  const float e = kEpsilon * 0.5f;
  const float g3 = 1.0f + 2.0f * ((3.0f * e) / (1.0f - 3.0f * e));
  const float pos[3] = {in_pos.x, in_pos.y, in_pos.z};
  const float dir[3] = {in_dir.x, in_dir.y, in_dir.z};
  t_min = 0.0f;
  t_max = max_t;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float t_near = (0.0f - pos[i]) / dir[i];
    float t_far = (1.0f - pos[i]) / dir[i];
    if (t_near > t_far) {
      float t = t_far;
      t_far = t_near;
      t_near = t;
    }
    t_far *= g3;
    t_min = t_near > t_min ? t_near : t_min;
    t_max = t_far < t_max ? t_far : t_max;
    if (t_min > t_max)
      return false;
  }
  return true;
}

ETX_DEV bool medium_intersects_bounds(const DMedium& m, const f3& in_pos, const f3& in_direction, float in_max_t, f3& medium_pos, f3& medium_dir, float& t_min, float& t_max) {  // :42-57
  if (in_max_t >= kMaxFloat)
    return false;
  const f3 extent = m.bounds_max - m.bounds_min;
  const f3 end_pos = in_pos + in_direction * in_max_t;
  const f3 medium_end_pos = (end_pos - m.bounds_min) / extent;  // BoundingBox::to_local, math.hxx:581-583
  medium_pos = (in_pos - m.bounds_min) / extent;
  const f3 delta = medium_end_pos - medium_pos;
  const float segment = length(delta);
  medium_dir = delta / segment;
  return medium_bounds(medium_pos, medium_dir, segment, t_min, t_max);
}

ETX_DEV float medium_sample_density(const DMedium& m, const f3& coord) {  // medium_sample_density_internal, :59-95
  if ((coord.x < 0.0f) || (coord.y < 0.0f) || (coord.z < 0.0f) || (coord.x >= 1.0f) || (coord.y >= 1.0f) || (coord.z >= 1.0f))
    return 0.0f;
  const float px = fminf(fmaxf(coord.x * float(m.dim_x) - 0.5f, 0.0f), float(m.dim_x) - 1.0f);
  const float py = fminf(fmaxf(coord.y * float(m.dim_y) - 0.5f, 0.0f), float(m.dim_y) - 1.0f);
  const float pz = fminf(fmaxf(coord.z * float(m.dim_z) - 0.5f, 0.0f), float(m.dim_z) - 1.0f);
  const uint32_t ix = min(m.dim_x - 1u, uint32_t(px)), nx = min(m.dim_x - 1u, ix + 1u);
  const uint32_t iy = min(m.dim_y - 1u, uint32_t(py)), ny = min(m.dim_y - 1u, iy + 1u);
  const uint32_t iz = min(m.dim_z - 1u, uint32_t(pz)), nz = min(m.dim_z - 1u, iz + 1u);
  const uint32_t sy = m.dim_x, sz = m.dim_x * m.dim_y;
  const float* density = m.density;
  const float d000 = density[ix + iy * sy + iz * sz], d001 = density[nx + iy * sy + iz * sz];
  const float d010 = density[ix + ny * sy + iz * sz], d011 = density[nx + ny * sy + iz * sz];
  const float d100 = density[ix + iy * sy + nz * sz], d101 = density[nx + iy * sy + nz * sz];
  const float d110 = density[ix + ny * sy + nz * sz], d111 = density[nx + ny * sy + nz * sz];
  const float dx = px - floorf(px), dy = py - floorf(py), dz = pz - floorf(pz);
  const float bottom = (d000 + (d001 - d000) * dx) + ((d010 + (d011 - d010) * dx) - (d000 + (d001 - d000) * dx)) * dy;
  const float top = (d100 + (d101 - d100) * dx) + ((d110 + (d111 - d110) * dx) - (d100 + (d101 - d100) * dx)) * dy;
  return bottom + (top - bottom) * dz;
}

// SpectralQuery::spectral_sample, spectrum.hxx:234-239
ETX_DEV float spectral_sample_wavelength(float rnd) {
  const float offset = 0x1.35ce7a0000000p-5f;
  const float scale = 1.0f - offset;
  return 538.0f - 138.888889f * atanhf(0.85691062f - 1.82750197f * (rnd * scale + offset));
}

struct Vtx {
  f3 pos, nrm, tan, btn;
  f2 tex;
};

// The shading record of a triangle (DScene::tri_shade, built on the device from the vertex / triangle tables whenever they are uploaded).
// A gather costs the vector memory pipeline about one cycle per ACTIVE LANE whatever its width (tools/micro/gather_bench.hip: dword,
// dwordx2 and dwordx4 loads at per-lane addresses all run at 0.5-0.9 lane-loads per clock and CU), and the shading kernels are bound by
// exactly that: the reference's layout (three indices, then three 56-byte vertices read as 3-float pieces) costs a shading point 18 loads
// and every later use of the triangle (the offset ray origins of the next segment and of each connection) 7 more. Here the same numbers sit
// in thirteen aligned 16-byte rows:
//   rows 0-2  position of vertex i, u of its texture coordinate       rows 3-5  normal of vertex i, v
//   row  6    geometric normal, material index (bits)                 rows 7-9  tangents, rows 10-12 bitangents
constexpr uint32_t kTriShadeStride = 13;

struct TriVerts {  // what shading_pos reads (rows 0-6). Keeping these 21 registers with the intersection was tried: an Isect of 180 bytes is no longer
                   // split into registers by the compiler (the whole struct went to scratch, 200 bytes per lane) - the rows are read again instead
  f3 p0, p1, p2, n0, n1, n2, geo_n;
};

struct Isect : public Vtx {
  f3 bc;
  uint32_t tri;
  f3 w_i;
  float t;
  uint32_t material;
  uint32_t emitter;
  f3 geo_n;  // the triangle's geometric normal (row 6 comes with the shading point)
};

// (Reading the rows through the constant address space, whose loads the compiler may re-use across the kernels' stores, changed nothing: the uses sit
// in different conditional blocks and are issued again either way - 697 vector loads in k_camera_shade<0,false> with and without.)
typedef const float4* TriRowsPtr;
ETX_DEV TriRowsPtr tri_rows(const DScene& s, const etx_abi_triangle& t) {
  return s.tri_shade + size_t(&t - s.triangles) * kTriShadeStride;
}
ETX_DEV f3 xyz(const float4& v) {
  return {v.x, v.y, v.z};
}

ETX_DEV TriVerts load_tri_verts(const DScene& s, const etx_abi_triangle& t) {
  const TriRowsPtr r = tri_rows(s, t);
  return {xyz(r[0]), xyz(r[1]), xyz(r[2]), xyz(r[3]), xyz(r[4]), xyz(r[5]), xyz(r[6])};
}

// scene.hxx:90-112 lerp_vertex: interpolate, re-orthogonalise (Gram-Schmidt), keep bitangent handedness
struct TriPoint {  // the interpolated vertex and what came with the triangle's rows (by value: out-pointers kept the caller's Isect in scratch)
  Vtx v;
  TriVerts tv;
  uint32_t material;
};

ETX_DEV TriPoint lerp_tri_point(const DScene& s, const etx_abi_triangle& t, const f3& bc) {
  const TriRowsPtr r = tri_rows(s, t);
  const float4 p0 = r[0], p1 = r[1], p2 = r[2], n0 = r[3], n1 = r[4], n2 = r[5], g = r[6];
  const float4 t0 = r[7], t1 = r[8], t2 = r[9], b0 = r[10], b1 = r[11], b2 = r[12];
  TriPoint out;
  out.tv = {xyz(p0), xyz(p1), xyz(p2), xyz(n0), xyz(n1), xyz(n2), xyz(g)};
  out.material = __float_as_uint(g.w);
  Vtx& v = out.v;
  v.pos = xyz(p0) * bc.x + xyz(p1) * bc.y + xyz(p2) * bc.z;
  v.nrm = xyz(n0) * bc.x + xyz(n1) * bc.y + xyz(n2) * bc.z;
  v.tan = xyz(t0) * bc.x + xyz(t1) * bc.y + xyz(t2) * bc.z;
  f3 b = xyz(b0) * bc.x + xyz(b1) * bc.y + xyz(b2) * bc.z;
  v.tex = {p0.w * bc.x + p1.w * bc.y + p2.w * bc.z, n0.w * bc.x + n1.w * bc.y + n2.w * bc.z};
  v.nrm = normalize(v.nrm);
  v.tan = normalize(v.tan - dot(v.tan, v.nrm) * v.nrm);
  f3 btn = cross(v.nrm, v.tan);
  v.btn = normalize(btn * (dot(btn, b) > 0.0f ? 1.0f : -1.0f));
  return out;
}

ETX_DEV Vtx lerp_vertex(const DScene& s, const etx_abi_triangle& t, const f3& bc) {
  return lerp_tri_point(s, t, bc).v;
}

ETX_DEV f3 lerp_pos(const DScene& s, const etx_abi_triangle& t, const f3& bc) {  // scene.hxx:77-81
  const TriRowsPtr r = tri_rows(s, t);
  return xyz(r[0]) * bc.x + xyz(r[1]) * bc.y + xyz(r[2]) * bc.z;
}
ETX_DEV f3 lerp_normal(const DScene& s, const etx_abi_triangle& t, const f3& bc) {  // scene.hxx:83-87
  const TriRowsPtr r = tri_rows(s, t);
  return normalize(xyz(r[3]) * bc.x + xyz(r[4]) * bc.y + xyz(r[5]) * bc.z);
}
ETX_DEV f2 lerp_uv(const DScene& s, const etx_abi_triangle& t, const f3& b) {  // scene.hxx:103-107
  const TriRowsPtr r = tri_rows(s, t);
  const float4 p0 = r[0], p1 = r[1], p2 = r[2], n0 = r[3], n1 = r[4], n2 = r[5];
  return {p0.w * b.x + p1.w * b.y + p2.w * b.z, n0.w * b.x + n1.w * b.y + n2.w * b.z};
}
// position, normal and texture coordinate of a point of a triangle from the six rows they share (area emitters: sample_emitter)
ETX_DEV void lerp_pos_normal_uv(const DScene& s, const etx_abi_triangle& t, const f3& b, f3& pos, f3& nrm, f2& uv) {
  const TriRowsPtr r = tri_rows(s, t);
  const float4 p0 = r[0], p1 = r[1], p2 = r[2], n0 = r[3], n1 = r[4], n2 = r[5];
  pos = xyz(p0) * b.x + xyz(p1) * b.y + xyz(p2) * b.z;
  nrm = normalize(xyz(n0) * b.x + xyz(n1) * b.y + xyz(n2) * b.z);
  uv = {p0.w * b.x + p1.w * b.y + p2.w * b.z, n0.w * b.x + n1.w * b.y + n2.w * b.z};
}

// scene.hxx:172-186 shading_pos: Phong-tessellation style origin (avoids the shadow terminator), then offset_ray
ETX_DEV f3 shading_pos(const TriVerts& tv, const f3& bc, const f3& w_o) {
  const f3 p0v = tv.p0, p1v = tv.p1, p2v = tv.p2, n0 = tv.n0, n1 = tv.n1, n2 = tv.n2;
  f3 geo_pos = p0v * bc.x + p1v * bc.y + p2v * bc.z;
  f3 sh_normal = normalize(n0 * bc.x + n1 * bc.y + n2 * bc.z);
  float direction = (dot(sh_normal, w_o) >= 0.0f) ? +1.0f : -1.0f;
  f3 d0 = direction * n0, d1 = direction * n1, d2 = direction * n2;
  f3 p0 = geo_pos - dot(geo_pos - p0v, d0) * d0;
  f3 p1 = geo_pos - dot(geo_pos - p1v, d1) * d1;
  f3 p2 = geo_pos - dot(geo_pos - p2v, d2) * d2;
  f3 sh_pos = p0 * bc.x + p1 * bc.y + p2 * bc.z;
  bool convex = dot(sh_pos - geo_pos, sh_normal) * direction > 0.0f;
  return offset_ray(convex ? sh_pos : geo_pos, tv.geo_n * direction);
}
ETX_DEV f3 shading_pos(const DScene& s, const etx_abi_triangle& t, const f3& bc, const f3& w_o) {
  return shading_pos(load_tri_verts(s, t), bc, w_o);
}

// ---------------------------------------------------------------------------------------------------------------
// images  sources/etx/render/shared/image.hxx

ETX_DEV float tex_coord(float u, float size, bool repeat) {  // image.hxx:163-178
  if (repeat) {
    float x = fmodf(u, size);
    return x < 0.0f ? (x + size) : x;
  }
  float hi = nextafterf(size, 0.0f);
  return u < 0.0f ? 0.0f : (u > hi ? hi : u);
}

struct ImageGather {
  float4 p00, p01, p10, p11;
};

ETX_DEV float4 scale4(const float4& a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
ETX_DEV float4 add4(const float4& a, const float4& b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

ETX_DEV ImageGather image_gather(const DImage& img, const f2 in_uv) {  // image.hxx:55-80
  float x0 = tex_coord(in_uv.x * img.fsize.x, img.fsize.x, (img.options & ETX_IMAGE_REPEAT_U) != 0);
  float y0 = tex_coord(in_uv.y * img.fsize.y, img.fsize.y, (img.options & ETX_IMAGE_REPEAT_V) != 0);
  float dx = x0 - floorf(x0);
  float dy = y0 - floorf(y0);
  uint32_t row_0 = min(static_cast<uint32_t>(y0), img.isize_y - 1u);
  uint32_t row_1 = min(row_0 + 1u, img.isize_y - 1u);
  uint32_t col_0 = min(static_cast<uint32_t>(x0), img.isize_x - 1u);
  uint32_t col_1 = min(col_0 + 1u, img.isize_x - 1u);
  uint32_t last = img.isize_x * img.isize_y - 1u;
  ImageGather g;
  g.p00 = scale4(img.pixels[min(col_0 + row_0 * img.isize_x, last)], (1.0f - dx) * (1.0f - dy));
  g.p01 = scale4(img.pixels[min(col_1 + row_0 * img.isize_x, last)], dx * (1.0f - dy));
  g.p10 = scale4(img.pixels[min(col_0 + row_1 * img.isize_x, last)], (1.0f - dx) * dy);
  g.p11 = scale4(img.pixels[min(col_1 + row_1 * img.isize_x, last)], dx * dy);
  return g;
}

ETX_DEV float4 image_evaluate(const DImage& img, const f2 uv, float* pdf) {  // image.hxx:82-96
  ImageGather g = image_gather(img, uv);
  if (pdf) {
    bool uniform = (img.options & ETX_IMAGE_UNIFORM_SAMPLING_TABLE) || (img.fsize.y == 1.0f);
    float s_t = uniform ? 1.0f : fmaxf(0.0f, sin_rev(0.5f * saturate(uv.y + 0.0f / img.fsize.y)));
    float t = luminance(mk3(add4(g.p00, g.p01))) * s_t;
    float s_b = uniform ? 1.0f : fmaxf(0.0f, sin_rev(0.5f * saturate(uv.y + 1.0f / img.fsize.y)));
    float b = luminance(mk3(add4(g.p10, g.p11))) * s_b;
    *pdf = (t + b) / img.normalization;
  }
  return add4(add4(g.p00, g.p01), add4(g.p10, g.p11));
}

// distribution.hxx:16-35 : binary search over the CDF
ETX_DEV uint32_t distribution_sample(const etx_abi_distribution_entry* values, uint32_t count, float rnd) {
  uint32_t b = 0, e = count;
  do {
    uint32_t m = b + (e - b) / 2;
    if (values[m].cdf >= rnd) {
      e = m;
    } else {
      b = m;
    }
  } while ((e - b) > 1);
  return b;
}

// image.hxx:126-161 : importance sample a texel, returns uv + pdf + value
ETX_DEV f2 image_sample(const DImage& img, const f2 rnd, float& image_pdf, float4& eval) {
  uint32_t ly = distribution_sample(img.y_entries, img.y_count, rnd.y);
  const etx_abi_distribution_entry* row = img.x_entries + size_t(ly) * img.x_stride;
  uint32_t lx = distribution_sample(row, img.x_stride, rnd.x);
  float x0c = row[lx].cdf, x1c = row[min(lx + 1u, img.x_stride - 1u)].cdf;
  float dx = rnd.x - x0c;
  if (x1c - x0c > 0.0f)
    dx /= (x1c - x0c);
  float y0c = img.y_entries[ly].cdf, y1c = img.y_entries[min(ly + 1u, img.y_count - 1u)].cdf;
  float dy = rnd.y - y0c;
  if (y1c - y0c > 0.0f)
    dy /= (y1c - y0c);
  f2 uv = {(float(lx) + dx) / img.fsize.x, (float(ly) + dy) / img.fsize.y};
  eval = image_evaluate(img, uv, &image_pdf);
  return uv;
}

// scene.hxx:250-320 : texture helpers (RGB mode: SpectralResponse = float3 `integrated`)
// rgb_response, spectrum.cxx:399-612: the weight of an RGB texel at one wavelength (table uploaded by the host)
ETX_DEV float rgb_response(const DScene& s, float wavelength, const f3& rgb) {
  if (luminance(rgb) == 0.0f)
    return 0.0f;
  const float last = s.rgb_response_first + float(s.rgb_response_count - 1u);
  if ((wavelength < s.rgb_response_first) || (wavelength > last))
    return 0.0f;
  const uint32_t wi = uint32_t(wavelength - s.rgb_response_first);
  const uint32_t wj = min(wi + 1u, s.rgb_response_count - 1u);
  const float dw = wavelength - floorf(wavelength);
  const float4 a = s.rgb_response[wi], b = s.rgb_response[wj];
  return rgb.x * (a.x + (b.x - a.x) * dw) + rgb.y * (a.y + (b.y - a.y) * dw) + rgb.z * (a.z + (b.z - a.z) * dw);
}

ETX_DEV f3 apply_image(const DScene& s, const etx_abi_spectral_image& img, const f2 uv, float* image_pdf, float wavelength) {  // scene.hxx:295-309
  if (image_pdf)
    *image_pdf = 0.0f;
  f3 result = spectrum_eval(s, img.spectrum_index, wavelength);
  if (img.image_index == kInvalid)
    return result;
  float4 e = image_evaluate(s.images[img.image_index], uv, image_pdf);
  if (s.spectral)  // apply_rgb, scene.hxx:249-260
    return result * rgb_response(s, wavelength, mk3(e));
  return result * mk3(e);
}
ETX_DEV float evaluate_image(const DScene& s, const etx_abi_sampled_image& img, const f2 uv, float default_value) {  // scene.hxx:272-281
  if ((img.image_index == kInvalid) || (img.channel >= 4u))
    return default_value;
  float4 e = image_evaluate(s.images[img.image_index], uv, nullptr);
  return img.channel == 0 ? e.x : (img.channel == 1 ? e.y : (img.channel == 2 ? e.z : e.w));
}
ETX_DEV f2 evaluate_roughness(const DScene& s, const etx_abi_material& m, const f2 uv) {  // scene.hxx:287-289
  float k = evaluate_image(s, m.roughness, uv, 1.0f);
  return {m.roughness.value.x * k, m.roughness.value.y * k};
}

// scene.hxx:188-200 orient_normals_to_hemisphere (normal mapping helper)
ETX_DEV f3 orient_normals_to_hemisphere(f3 n_s, const f3& n_g, const f3& v) {
  const float i_dot_g = dot(v, n_g);
  float i_dot_s = dot(v, n_s);
  for (uint32_t i = 0; ((i_dot_s * i_dot_g) <= kEpsilon) && (i < 16u); ++i) {
    n_s = normalize(8.0f * n_s + n_g);
    i_dot_s = dot(v, n_s);
  }
  return n_s;
}

// scene.hxx:202-226 make_intersection: expand the 16-byte hit record (u, v, t, triangle) into a shading point
ETX_DEV Isect make_intersection(const DScene& s, const f3& w_i, float u, float v, float t, uint32_t tri_index) {
  f3 bc = barycentrics(u, v);
  const etx_abi_triangle& tri = s.triangles[tri_index];
  const TriPoint point = lerp_tri_point(s, tri, bc);
  Isect r;
  static_cast<Vtx&>(r) = point.v;
  r.geo_n = point.tv.geo_n;
  r.material = point.material;
  r.bc = bc;
  r.tri = tri_index;
  r.w_i = w_i;
  r.t = t;
  r.emitter = s.triangle_to_emitter[tri_index];
  if (s.normal_mapped_materials == 0u)  // scene-uniform: the two loads below are not issued at all
    return r;
  const etx_abi_material& mat = s.materials[r.material];
  if ((mat.normal_image_index != kInvalid) && (mat.normal_scale > kEpsilon)) {
    float4 value = image_evaluate(s.images[mat.normal_image_index], r.tex, nullptr);  // image.hxx:117-124 evaluate_normal
    float sc = mat.normal_scale;
    f3 sn = {sc * (value.x * 2.0f - 1.0f), sc * (value.y * 2.0f - 1.0f), sc * (value.z * 2.0f - 1.0f) + (1.0f - sc)};
    r.nrm = normalize(r.tan * sn.x + r.btn * sn.y + r.nrm * sn.z);
    r.nrm = orient_normals_to_hemisphere(r.nrm, r.geo_n, w_i);
    r.tan = normalize(r.tan - dot(r.tan, r.nrm) * r.nrm);
    r.btn = normalize(cross(r.nrm, r.tan));
  }
  return r;
}

}  // namespace etxd
