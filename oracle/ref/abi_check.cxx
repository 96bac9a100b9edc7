// ORACLE BUILD RECIPE - compile-time proof that include/etx_scene_abi.h restates the reference's POD layouts.
// Includes the reference headers (by path, from /root/reference) next to the backend's C mirror and
// static_asserts size + field offsets. Compiled by oracle/build_ref.sh; produces no code.
#include <etx/core/core.hxx>
#include <etx/render/host/film.hxx>
#include <etx/rt/integrators/vcm_cpu.hxx>
#include <etx/rt/shared/vcm_shared.hxx>
#include <cstddef>

#include "../../include/etx_scene_abi.h"

#define SAME_SIZE(REF, ABI) static_assert(sizeof(REF) == sizeof(ABI), "size mismatch: " #REF " vs " #ABI)
#define SAME_FIELD(REF, RF, ABI, AF) static_assert(offsetof(REF, RF) == offsetof(ABI, AF), "offset mismatch: " #REF "::" #RF)

using namespace etx;

SAME_SIZE(Vertex, etx_abi_vertex);
SAME_FIELD(Vertex, nrm, etx_abi_vertex, nrm);
SAME_FIELD(Vertex, tan, etx_abi_vertex, tan);
SAME_FIELD(Vertex, btn, etx_abi_vertex, btn);
SAME_FIELD(Vertex, tex, etx_abi_vertex, tex);

SAME_SIZE(Triangle, etx_abi_triangle);
SAME_FIELD(Triangle, material_index, etx_abi_triangle, material_index);
SAME_FIELD(Triangle, geo_n, etx_abi_triangle, geo_n);

SAME_SIZE(Material, etx_abi_material);
SAME_FIELD(Material, reflectance, etx_abi_material, reflectance);
SAME_FIELD(Material, scattering, etx_abi_material, scattering);
SAME_FIELD(Material, emission, etx_abi_material, emission);
SAME_FIELD(Material, roughness, etx_abi_material, roughness);
SAME_FIELD(Material, metalness, etx_abi_material, metalness);
SAME_FIELD(Material, transmission, etx_abi_material, transmission);
SAME_FIELD(Material, subsurface, etx_abi_material, subsurface);
SAME_FIELD(Material, thinfilm, etx_abi_material, thinfilm);
SAME_FIELD(Material, ext_ior, etx_abi_material, ext_ior);
SAME_FIELD(Material, int_ior, etx_abi_material, int_ior);
SAME_FIELD(Material, cls, etx_abi_material, cls);
SAME_FIELD(Material, int_medium, etx_abi_material, int_medium);
SAME_FIELD(Material, ext_medium, etx_abi_material, ext_medium);
SAME_FIELD(Material, normal_image_index, etx_abi_material, normal_image_index);
SAME_FIELD(Material, diffuse_variation, etx_abi_material, diffuse_variation);
SAME_FIELD(Material, two_sided, etx_abi_material, two_sided);
SAME_FIELD(Material, normal_scale, etx_abi_material, normal_scale);
SAME_FIELD(Material, opacity, etx_abi_material, opacity);
SAME_FIELD(Material, emission_collimation, etx_abi_material, emission_collimation);
SAME_SIZE(SampledImage, etx_abi_sampled_image);
SAME_SIZE(Thinfilm, etx_abi_thinfilm);
SAME_FIELD(Thinfilm, thinkness_image, etx_abi_thinfilm, thickness_image);
SAME_SIZE(SubsurfaceMaterial, etx_abi_subsurface);
SAME_FIELD(SubsurfaceMaterial, cls, etx_abi_subsurface, cls);
static_assert(uint32_t(Material::Class::Diffuse) == ETX_MAT_DIFFUSE && uint32_t(Material::Class::Conductor) == ETX_MAT_CONDUCTOR &&
              uint32_t(Material::Class::Boundary) == ETX_MAT_BOUNDARY && uint32_t(Material::Class::Void) == ETX_MAT_VOID &&
              uint32_t(Material::Class::Count) == ETX_MAT_COUNT && uint32_t(Material::Class::Dielectric) == ETX_MAT_DIELECTRIC &&
              uint32_t(Material::Class::Mirror) == ETX_MAT_MIRROR && uint32_t(Material::Class::Principled) == ETX_MAT_PRINCIPLED);

SAME_SIZE(EmitterProfile, etx_abi_emitter_profile);
SAME_FIELD(EmitterProfile, direction, etx_abi_emitter_profile, direction);
SAME_FIELD(EmitterProfile, cls, etx_abi_emitter_profile, cls);
SAME_FIELD(EmitterProfile, angular_size, etx_abi_emitter_profile, angular_size);
SAME_FIELD(EmitterProfile, equivalent_disk_size, etx_abi_emitter_profile, equivalent_disk_size);
SAME_FIELD(EmitterProfile, angular_size_cosine, etx_abi_emitter_profile, angular_size_cosine);
static_assert(uint32_t(EmitterProfile::Class::Area) == ETX_EMITTER_AREA && uint32_t(EmitterProfile::Class::Environment) == ETX_EMITTER_ENVIRONMENT &&
              uint32_t(EmitterProfile::Class::Directional) == ETX_EMITTER_DIRECTIONAL);

SAME_SIZE(Emitter, etx_abi_emitter);
SAME_FIELD(Emitter, profile, etx_abi_emitter, profile);
SAME_FIELD(Emitter, triangle_index, etx_abi_emitter, triangle_index);
SAME_FIELD(Emitter, spectrum_weight, etx_abi_emitter, spectrum_weight);
SAME_FIELD(Emitter, additional_weight, etx_abi_emitter, additional_weight);
SAME_FIELD(Emitter, triangle_area, etx_abi_emitter, triangle_area);

SAME_SIZE(Distribution, etx_abi_distribution);
SAME_SIZE(Distribution::Entry, etx_abi_distribution_entry);
SAME_FIELD(Distribution, total_weight, etx_abi_distribution, total_weight);

SAME_SIZE(Image, etx_abi_image);
SAME_FIELD(Image, x_distributions, etx_abi_image, x_distributions);
SAME_FIELD(Image, y_distribution, etx_abi_image, y_distribution);
SAME_FIELD(Image, fsize, etx_abi_image, fsize);
SAME_FIELD(Image, offset, etx_abi_image, offset);
SAME_FIELD(Image, scale, etx_abi_image, scale);
SAME_FIELD(Image, isize, etx_abi_image, isize);
SAME_FIELD(Image, normalization, etx_abi_image, normalization);
SAME_FIELD(Image, options, etx_abi_image, options);
SAME_FIELD(Image, format, etx_abi_image, format);
static_assert(uint32_t(Image::Format::RGBA32F) == ETX_IMAGE_FORMAT_RGBA32F && uint32_t(Image::Format::RGBA8) == ETX_IMAGE_FORMAT_RGBA8);
static_assert(Image::RepeatU == ETX_IMAGE_REPEAT_U && Image::RepeatV == ETX_IMAGE_REPEAT_V && Image::HasAlphaChannel == ETX_IMAGE_HAS_ALPHA &&
              Image::UniformSamplingTable == ETX_IMAGE_UNIFORM_SAMPLING_TABLE);

SAME_SIZE(Medium, etx_abi_medium);
SAME_FIELD(Medium, bounds, etx_abi_medium, bounds_min);
SAME_FIELD(Medium, cls, etx_abi_medium, cls);
SAME_FIELD(Medium, enable_explicit_connections, etx_abi_medium, enable_explicit_connections);
SAME_FIELD(Medium, absorption_index, etx_abi_medium, absorption_index);
SAME_FIELD(Medium, scattering_index, etx_abi_medium, scattering_index);
SAME_FIELD(Medium, phase_function_g, etx_abi_medium, phase_function_g);
SAME_FIELD(Medium, max_sigma, etx_abi_medium, max_sigma);
SAME_FIELD(Medium, dimensions, etx_abi_medium, dimensions);

SAME_SIZE(SpectralDistribution, etx_abi_spectrum);
SAME_FIELD(SpectralDistribution, spectral_entry_count, etx_abi_spectrum, entry_count);
static_assert(offsetof(etx_abi_spectrum, integrated) == offsetof(SpectralDistribution, spectral_entry_count) + sizeof(uint32_t));
static_assert(spectrum::WavelengthCount == ETX_ABI_SPECTRUM_MAX_ENTRIES);

SAME_SIZE(Camera, etx_abi_camera);
SAME_FIELD(Camera, position, etx_abi_camera, position);
SAME_FIELD(Camera, cls, etx_abi_camera, cls);
SAME_FIELD(Camera, tan_half_fov, etx_abi_camera, tan_half_fov);
SAME_FIELD(Camera, side, etx_abi_camera, side);
SAME_FIELD(Camera, aspect, etx_abi_camera, aspect);
SAME_FIELD(Camera, up, etx_abi_camera, up);
SAME_FIELD(Camera, area, etx_abi_camera, area);
SAME_FIELD(Camera, direction, etx_abi_camera, direction);
SAME_FIELD(Camera, image_plane, etx_abi_camera, image_plane);
SAME_FIELD(Camera, film_size, etx_abi_camera, film_size);
SAME_FIELD(Camera, lens_radius, etx_abi_camera, lens_radius);
SAME_FIELD(Camera, focal_distance, etx_abi_camera, focal_distance);
SAME_FIELD(Camera, clip_near, etx_abi_camera, clip_near);
SAME_FIELD(Camera, clip_far, etx_abi_camera, clip_far);
SAME_FIELD(Camera, lens_image, etx_abi_camera, lens_image);
SAME_FIELD(Camera, medium_index, etx_abi_camera, medium_index);

SAME_SIZE(Scene, etx_abi_scene);
SAME_FIELD(Scene, triangles, etx_abi_scene, triangles);
SAME_FIELD(Scene, triangle_to_emitter, etx_abi_scene, triangle_to_emitter);
SAME_FIELD(Scene, materials, etx_abi_scene, materials);
SAME_FIELD(Scene, emitter_profiles, etx_abi_scene, emitter_profiles);
SAME_FIELD(Scene, emitter_instances, etx_abi_scene, emitter_instances);
SAME_FIELD(Scene, images, etx_abi_scene, images);
SAME_FIELD(Scene, mediums, etx_abi_scene, mediums);
SAME_FIELD(Scene, spectrums, etx_abi_scene, spectrums);
SAME_FIELD(Scene, emitters_distribution, etx_abi_scene, emitters_distribution);
SAME_FIELD(Scene, environment_emitters, etx_abi_scene, environment_emitters);
SAME_FIELD(Scene, bounding_sphere_center, etx_abi_scene, bounding_sphere_center);
SAME_FIELD(Scene, bounding_sphere_radius, etx_abi_scene, bounding_sphere_radius);
SAME_FIELD(Scene, pixel_sampler, etx_abi_scene, pixel_sampler);
SAME_FIELD(Scene, min_path_length, etx_abi_scene, min_path_length);
SAME_FIELD(Scene, max_path_length, etx_abi_scene, max_path_length);
SAME_FIELD(Scene, samples, etx_abi_scene, samples);
SAME_FIELD(Scene, random_path_termination, etx_abi_scene, random_path_termination);
SAME_FIELD(Scene, radiance_clamp, etx_abi_scene, radiance_clamp);
SAME_FIELD(Scene, black_spectrum, etx_abi_scene, black_spectrum);
SAME_FIELD(Scene, subsurface_scatter_material, etx_abi_scene, subsurface_scatter_material);
SAME_FIELD(Scene, subsurface_exit_material, etx_abi_scene, subsurface_exit_material);
SAME_FIELD(Scene, default_dielectric_eta, etx_abi_scene, default_dielectric_eta);
SAME_FIELD(Scene, flags, etx_abi_scene, flags);
static_assert(Scene::Spectral == ETX_SCENE_SPECTRAL && Scene::Committed == ETX_SCENE_COMMITTED);
static_assert(EnvironmentEmitters::kMaxCount == ETX_ABI_MAX_ENVIRONMENT_EMITTERS);

SAME_SIZE(VCMOptions, etx_abi_vcm_options);
SAME_FIELD(VCMOptions, radius_decay, etx_abi_vcm_options, radius_decay);
SAME_FIELD(VCMOptions, kernel, etx_abi_vcm_options, kernel);
SAME_FIELD(VCMOptions, initial_radius, etx_abi_vcm_options, initial_radius);
SAME_FIELD(VCMOptions, blue_noise, etx_abi_vcm_options, blue_noise);
static_assert(VCMOptions::FullOptions == ETX_VCM_FULL_OPTIONS && VCMOptions::EnableMerging == ETX_VCM_ENABLE_MERGING && VCMOptions::MergeVertices == ETX_VCM_MERGE_VERTICES);

SAME_SIZE(PTOptions, etx_abi_pt_options);
SAME_FIELD(PTOptions, nee, etx_abi_pt_options, nee);
SAME_FIELD(PTOptions, direct, etx_abi_pt_options, direct);
SAME_FIELD(PTOptions, mis, etx_abi_pt_options, mis);
SAME_FIELD(PTOptions, blue_noise, etx_abi_pt_options, blue_noise);

int etx_abi_check_anchor = 0;
