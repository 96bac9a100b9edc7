// ORACLE / TEST INFRASTRUCTURE ONLY - never linked into libetx_hip.so.
//
// Headless driver around the reference's own host code, compiled read-only from /root/reference:
//   SceneRepresentation::load_from_file  (sources/etx/render/host/scene_representation.cxx:679-838)
//   Raytracing::commit_changes           (sources/etx/rt/rt.cxx:58-64; here: oracle/shims/raytracing_bvh.cxx)
//   CPUPathTracing / CPUVCM ::run/update (sources/etx/rt/integrators/path_tracing.cxx:85-110, vcm_cpu.cxx:255-276)
//   Film::layer                          (sources/etx/render/host/film.cxx:381-418)
// It replaces the GUI pump of sources/raytracer/app.cxx:126-159 with a loop over Integrator::update() until
// State::Stopped, then writes the film layers as raw float4 and (optionally) a byte-exact snapshot of the loaded
// etx::Scene / etx::Camera (the input ABI of the HIP backend) so that GPU-side tests and bench.py
// can run where /root/reference does not exist.
//
// Output formats (little endian):
//   film  : "ETXFILM1" u32 width u32 height u32 layers u32 spp f64 seconds u32 threads u32 pad ; layers * (w*h float4)
//           layer order: Film::CameraImage, Film::LightImage, Film::Result
//   scene : see write_snapshot() below ("ETXSCENE1")
#include <etx/core/core.hxx>
#include <etx/core/environment.hxx>
#include <etx/render/host/film.hxx>
#include <etx/render/host/scene_representation.hxx>
#include <etx/render/shared/ior_database.hxx>
// Two targets from this file (oracle/build_ref.sh): etx_oracle = the checker, with the reference's CPU integrators next to the HIP binding;
// etx_hip_render (-DETX_DRIVER_HIP_ONLY) = the headless host of the HIP backend alone: scene loader, film and the
// binding of integration/etx_hip_integrators.hxx, no CPU integrator linked.
#if !defined(ETX_DRIVER_HIP_ONLY)
#include <etx/rt/integrators/path_tracing.hxx>
#include <etx/rt/integrators/vcm_cpu.hxx>
#include <etx/rt/integrators/bidirectional.hxx>
#else
#include <etx/rt/integrators/integrator.hxx>
#endif
#include <etx/rt/shared/vcm_shared.hxx>
#include <bluenoise.hxx>

#include <etx_hip_integrators.hxx>  // integration/: the reference-side binding of libetx_hip.so (HIPVCM, HIPPathTracing)
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace etx {
extern std::atomic<uint64_t> g_oracle_rays_trace;
extern std::atomic<uint64_t> g_oracle_rays_transmittance;
extern std::atomic<uint64_t> g_oracle_rays_material;
}  // namespace etx

using namespace etx;

namespace {

struct Blob {
  std::vector<uint8_t> bytes;

  uint64_t append(const void* p, uint64_t size) {
    uint64_t offset = (bytes.size() + 15u) & ~uint64_t(15);
    bytes.resize(offset + size);
    if (size > 0)
      memcpy(bytes.data() + offset, p, size);
    return offset;
  }
};

// Snapshot layout:
//   "ETXSCENE1" (16 bytes, zero padded) | u64 scene_offset | u64 camera_offset | u64 fixup_count | u64 total_size
//   fixups[fixup_count] : u64 offset-of-pointer-field (the field holds a blob offset, 0 = null) |
//   payload (16-byte aligned chunks: raw Scene bytes, raw Camera bytes, every array an ArrayView points to)
// Loading = read file, add the base address to every listed pointer field.
struct Snapshot {
  Blob blob;
  std::vector<uint64_t> fixups;

  template <class T>
  void patch(uint64_t struct_offset, const ArrayView<T>& view_in_struct, const void* struct_base, uint64_t data_offset) {
    uint64_t field = struct_offset + uint64_t(reinterpret_cast<const uint8_t*>(&view_in_struct.a) - reinterpret_cast<const uint8_t*>(struct_base));
    uint64_t value = view_in_struct.count ? data_offset : 0;
    memcpy(blob.bytes.data() + field, &value, sizeof(uint64_t));
    fixups.push_back(field);
  }

  template <class T>
  uint64_t store_array(uint64_t struct_offset, const ArrayView<T>& view, const void* struct_base) {
    uint64_t off = blob.append(view.a, view.count * sizeof(T));
    patch(struct_offset, view, struct_base, off);
    return off;
  }

  void store_distribution(uint64_t struct_offset, const Distribution& d, const void* struct_base) {
    store_array(struct_offset, d.values, struct_base);
  }
};

bool write_snapshot(const char* path, const Scene& scene, const Camera& camera) {
  Snapshot s;
  s.blob.bytes.reserve(1u << 20);
  s.blob.bytes.resize(64);  // header placeholder

  uint64_t scene_off = s.blob.append(&scene, sizeof(Scene));
  uint64_t camera_off = s.blob.append(&camera, sizeof(Camera));

  s.store_array(scene_off, scene.vertices, &scene);
  s.store_array(scene_off, scene.triangles, &scene);
  s.store_array(scene_off, scene.triangle_to_emitter, &scene);
  s.store_array(scene_off, scene.materials, &scene);
  s.store_array(scene_off, scene.emitter_profiles, &scene);
  s.store_array(scene_off, scene.emitter_instances, &scene);
  s.store_array(scene_off, scene.spectrums, &scene);
  s.store_distribution(scene_off, scene.emitters_distribution, &scene);

  uint64_t images_off = s.store_array(scene_off, scene.images, &scene);
  for (uint64_t i = 0; i < scene.images.count; ++i) {
    const Image& img = scene.images[i];
    uint64_t img_off = images_off + i * sizeof(Image);
    if (img.format == Image::Format::RGBA8) {
      s.store_array(img_off, img.pixels.u8, &img);
    } else {
      s.store_array(img_off, img.pixels.f32, &img);
    }
    s.store_distribution(img_off, img.y_distribution, &img);
    uint64_t xd_off = s.store_array(img_off, img.x_distributions, &img);
    for (uint64_t j = 0; j < img.x_distributions.count; ++j) {
      s.store_distribution(xd_off + j * sizeof(Distribution), img.x_distributions[j], &img.x_distributions[j]);
    }
  }

  uint64_t mediums_off = s.store_array(scene_off, scene.mediums, &scene);
  for (uint64_t i = 0; i < scene.mediums.count; ++i) {
    const Medium& m = scene.mediums[i];
    s.store_array(mediums_off + i * sizeof(Medium), m.density, &m);
  }

  uint64_t fix_off = s.blob.append(s.fixups.data(), s.fixups.size() * sizeof(uint64_t));
  uint64_t total = s.blob.bytes.size();

  uint8_t header[64] = {};
  memcpy(header, "ETXSCENE1", 9);
  uint64_t fields[6] = {scene_off, camera_off, uint64_t(s.fixups.size()), fix_off, total, (uint64_t(sizeof(Scene)) << 32) | uint64_t(sizeof(Camera))};
  memcpy(header + 16, fields, sizeof(fields));
  memcpy(s.blob.bytes.data(), header, sizeof(header));

  FILE* f = fopen(path, "wb");
  if (f == nullptr)
    return false;
  fwrite(s.blob.bytes.data(), 1, s.blob.bytes.size(), f);
  fclose(f);
  printf("snapshot: %s (%llu bytes, %llu pointer fixups)\n", path, (unsigned long long)total, (unsigned long long)s.fixups.size());
  return true;
}

bool write_film(const char* path, Film& film, uint32_t spp, double seconds, uint32_t threads) {
  FILE* f = fopen(path, "wb");
  if (f == nullptr)
    return false;
  uint2 dim = film.size();
  char magic[8] = {'E', 'T', 'X', 'F', 'I', 'L', 'M', '1'};
  uint32_t head[4] = {dim.x, dim.y, 5u, spp};
  uint32_t tail[2] = {threads, 0u};
  fwrite(magic, 1, 8, f);
  fwrite(head, sizeof(uint32_t), 4, f);
  fwrite(&seconds, sizeof(double), 1, f);
  fwrite(tail, sizeof(uint32_t), 2, f);
  const uint32_t layers[5] = {Film::CameraImage, Film::LightImage, Film::Result, Film::Normals, Film::Albedo};
  for (uint32_t l : layers) {
    const float4* data = film.layer(l);
    fwrite(data, sizeof(float4), size_t(dim.x) * dim.y, f);
  }
  fclose(f);
  return true;
}

// Inverse of write_snapshot: returns pointers into `storage`.
bool load_snapshot(const char* path, std::vector<uint8_t>& storage, const Scene*& scene, const Camera*& camera) {
  if (load_binary_file(path, storage) == false)
    return false;
  if ((storage.size() < 64) || (memcmp(storage.data(), "ETXSCENE1", 9) != 0))
    return false;
  uint64_t fields[6] = {};
  memcpy(fields, storage.data() + 16, sizeof(fields));
  if ((fields[4] != storage.size()) || (fields[5] != ((uint64_t(sizeof(Scene)) << 32) | uint64_t(sizeof(Camera)))))
    return false;
  uint64_t base = reinterpret_cast<uint64_t>(storage.data());
  for (uint64_t i = 0; i < fields[2]; ++i) {
    uint64_t field_offset = 0;
    memcpy(&field_offset, storage.data() + fields[3] + i * sizeof(uint64_t), sizeof(uint64_t));
    uint64_t value = 0;
    memcpy(&value, storage.data() + field_offset, sizeof(uint64_t));
    value = value ? value + base : 0;
    memcpy(storage.data() + field_offset, &value, sizeof(uint64_t));
  }
  scene = reinterpret_cast<const Scene*>(storage.data() + fields[0]);
  camera = reinterpret_cast<const Camera*>(storage.data() + fields[1]);
  return true;
}

#if !defined(ETX_DRIVER_HIP_ONLY)
// Known-answer vectors straight from the reference's headers (pins oracle/kat.c and the device KAT kernels).
int print_kat() {
  printf("{\n");
  printf("  \"sampler\": [");
  const uint32_t ab[][2] = {{1u, 2u}, {0u, 0u}, {12345u, 7u}, {2073599u, 63u}, {0xffffffffu, 1u}, {16383u, 1023u}};
  for (uint32_t i = 0; i < 6; ++i) {
    Sampler s(ab[i][0], ab[i][1]);
    uint32_t seed = s.seed;
    float a = s.next(), b = s.next(), c = s.next();
    printf("%s[%u, %u, %u, %.9g, %.9g, %.9g]", i ? ", " : "", ab[i][0], ab[i][1], seed, a, b, c);
  }
  printf("],\n  \"offset_ray\": [");
  const float pn[][6] = {{0.5f, 1.0f, -2.0f, 0.0f, 1.0f, 0.0f}, {0.01f, -0.02f, 0.03f, 0.57735f, 0.57735f, 0.57735f}, {-1.0f, 0.5f, 0.25f, 1.0f, 0.0f, 0.0f}, {100.0f, -50.0f, 0.001f, -0.6f, 0.0f, 0.8f}};
  for (uint32_t i = 0; i < 4; ++i) {
    float3 r = offset_ray({pn[i][0], pn[i][1], pn[i][2]}, {pn[i][3], pn[i][4], pn[i][5]});
    printf("%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", pn[i][0], pn[i][1], pn[i][2], pn[i][3], pn[i][4], pn[i][5], r.x, r.y, r.z);
  }
  printf("],\n  \"orthonormal_basis\": [");
  const float nn[][3] = {{0.0f, 1.0f, 0.0f}, {0.57735026f, 0.57735026f, 0.57735026f}, {1.0f, 0.0f, 0.0f}, {-0.6f, 0.0f, 0.8f}, {0.0f, 0.0f, -1.0f}};
  for (uint32_t i = 0; i < 5; ++i) {
    auto b = orthonormal_basis({nn[i][0], nn[i][1], nn[i][2]});
    printf("%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", nn[i][0], nn[i][1], nn[i][2], b.u.x, b.u.y, b.u.z, b.v.x, b.v.y, b.v.z);
  }
  printf("],\n  \"sample_cosine\": [");
  const float rc[][5] = {{0.25f, 0.5f, 0.0f, 1.0f, 0.0f}, {0.9f, 0.1f, 1.0f, 0.0f, 0.0f}, {0.01f, 0.77f, -0.6f, 0.0f, 0.8f}, {0.5f, 0.999f, 0.0f, 0.0f, -1.0f}};
  for (uint32_t i = 0; i < 4; ++i) {
    float3 r = sample_cosine_distribution(float2{rc[i][0], rc[i][1]}, float3{rc[i][2], rc[i][3], rc[i][4]}, 1.0f);
    printf("%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", rc[i][0], rc[i][1], rc[i][2], rc[i][3], rc[i][4], r.x, r.y, r.z);
  }
  printf("],\n  \"cell_index\": [");
  const int32_t cells[][4] = {{0, 0, 0, 1023}, {1, 2, 3, 1023}, {-1, 5, 100, 65535}, {12345, -678, 9, 16777215}, {-2147483647, 2147483647, 1, 255}};
  VCMSpatialGridData grid = {};
  for (uint32_t i = 0; i < 5; ++i) {
    grid.hash_table_mask = uint32_t(cells[i][3]);
    printf("%s[%d, %d, %d, %u, %u]", i ? ", " : "", cells[i][0], cells[i][1], cells[i][2], uint32_t(cells[i][3]), grid.cell_index(cells[i][0], cells[i][1], cells[i][2]));
  }
  printf("],\n  \"sample_disk\": [");
  const float rd[][2] = {{0.5f, 0.5f}, {0.1f, 0.9f}, {0.75f, 0.25f}, {0.0f, 1.0f}, {0.3f, 0.31f}};
  for (uint32_t i = 0; i < 5; ++i) {
    float2 r = sample_disk({rd[i][0], rd[i][1]});
    printf("%s[%.9g, %.9g, %.9g, %.9g]", i ? ", " : "", rd[i][0], rd[i][1], r.x, r.y);
  }
  // sample_blue_noise(pixel, total_samples = 64, current_sample, dimension 0 / 2 / 4) as vcm_camera_step calls it
  printf("],\n  \"blue_noise_64spp\": [");
  const uint32_t bn[][3] = {{0u, 0u, 0u}, {1u, 0u, 0u}, {0u, 1u, 0u}, {5u, 7u, 3u}, {127u, 127u, 63u}, {128u, 130u, 1u}, {1919u, 1079u, 17u}, {777u, 345u, 255u}, {64u, 64u, 200u}};
  for (uint32_t i = 0; i < 9; ++i) {
    float2 a = sample_blue_noise({bn[i][0], bn[i][1]}, 64u, bn[i][2], 0u);
    float2 b = sample_blue_noise({bn[i][0], bn[i][1]}, 64u, bn[i][2], 2u);
    float2 c = sample_blue_noise({bn[i][0], bn[i][1]}, 64u, bn[i][2], 4u);
    printf("%s[%u, %u, %u, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", bn[i][0], bn[i][1], bn[i][2], a.x, a.y, b.x, b.y, c.x, c.y);
  }
  printf("]\n}\n");
  return 0;
}
#endif

void usage() {
  printf(
    "etx_oracle --scene file.json | --load-snapshot scene.bin --integrator pt|vcm|bdpt|hip-vcm|hip-pt|hip-bdpt [--spp N] [--out film.raw] [--snapshot scene.bin]\n"
    "           [--data /root/reference/bin/] [--opt key=value]... [--max-iterations N] [--pixel-size N] [--inject-density N]\n"
    "           hip-* only: [--checkpoint file] (film state when the render stops) [--resume file] (continue that render)\n");
}

}  // namespace

int main(int argc, char** argv) {
  std::string load_snapshot_file, checkpoint_file, resume_file;
  std::string scene_file, integrator_name = "vcm", out_file, snapshot_file, data_folder = "/root/reference/bin/";
  std::vector<std::pair<std::string, std::string>> opts;
  int64_t spp = -1;
  int64_t max_iterations = -1;
  std::string adaptive_log;  // --adaptive-log: pixels Film::active_pixel reports as still sampled, after every completed iteration
  float noise_threshold = -1.0f;  // < 0: keep the scene value
  int64_t subsurface_class = -1;  // >= 1: every subsurface material of the loaded scene gets this SubsurfaceMaterial::Class (1 random walk, 2 Christensen-Burley)
  uint32_t pixel_size = 0;      // >= 1: Film::set_pixel_size before the render (the GUI's preview while the camera moves, app.cxx:135)
  float probe_ray[7] = {};
  uint32_t probe_material = kInvalidIndex, probe_max_hits = 0;  // --probe-continuous-trace
  uint32_t inject_density = 0;  // N: every medium of the loaded scene becomes Heterogeneous with a procedural N^3 density grid
  for (int i = 1; i < argc; ++i) {
    auto next = [&]() -> const char* {
      return (i + 1 < argc) ? argv[++i] : "";
    };
    if (strcmp(argv[i], "--scene") == 0)
      scene_file = next();
    else if (strcmp(argv[i], "--load-snapshot") == 0)
      load_snapshot_file = next();
#if !defined(ETX_DRIVER_HIP_ONLY)
    else if (strcmp(argv[i], "--kat") == 0)
      return print_kat();
#endif
    else if (strcmp(argv[i], "--dump-cie") == 0) {
      // --dump-cie <file>: first wavelength, count, then spectrum::spectral_xyz(i) (the observer behind SpectralResponse::to_xyz)
      const char* file = next();
      FILE* f = fopen(file, "wb");
      if (f == nullptr)
        return 2;
      float head[2] = {spectrum::kShortestWavelength, float(spectrum::WavelengthCount)};
      fwrite(head, sizeof(float), 2, f);
      for (uint32_t k = 0; k < spectrum::WavelengthCount; ++k) {
        float3 v = spectrum::spectral_xyz(k);
        fwrite(&v.x, sizeof(float), 3, f);
      }
      fclose(f);
      return 0;
    }
    else if (strcmp(argv[i], "--dump-rgb-response") == 0) {
      // --dump-rgb-response <file>: first wavelength, count, then rgb_response (spectrum.cxx:399-612) of the unit colours at every
      // integer wavelength = the rows of its table (etx_hip_upload_rgb_response)
      const char* file = next();
      FILE* f = fopen(file, "wb");
      if (f == nullptr)
        return 2;
      float head[2] = {spectrum::kRGBResponseShortestWavelength, float(spectrum::RGBResponseWavelengthCount)};
      fwrite(head, sizeof(float), 2, f);
      for (uint32_t k = 0; k < spectrum::RGBResponseWavelengthCount; ++k) {
        const SpectralQuery q = {spectrum::kRGBResponseShortestWavelength + float(k), SpectralQuery::Spectral};
        float row[3] = {rgb_response(q, {1.0f, 0.0f, 0.0f}).value, rgb_response(q, {0.0f, 1.0f, 0.0f}).value, rgb_response(q, {0.0f, 0.0f, 1.0f}).value};
        fwrite(row, sizeof(float), 3, f);
      }
      fclose(f);
      return 0;
    }
    else if (strcmp(argv[i], "--dump-bluenoise") == 0) {
      // --dump-bluenoise <samples> <file>: what sample_blue_noise (path_tracing.cxx:173-178) returns for this sample-count
      // class, as bytes: value[(((py * 128 + px) * 256) + sample) * 8 + dimension], float = (0.5 + value) / 256
      const uint32_t samples = uint32_t(atoll(next()));
      const char* file = next();
      std::vector<uint8_t> table(size_t(128) * 128 * 256 * 8);
      for (uint32_t py = 0; py < 128; ++py)
        for (uint32_t px = 0; px < 128; ++px)
          for (uint32_t sample = 0; sample < 256; ++sample) {
            BNSampler smp(px, py, samples, sample);
            for (uint32_t d = 0; d < 8; ++d)
              table[(((size_t(py) * 128 + px) * 256) + sample) * 8 + d] = uint8_t(smp.get(d) * 256.0f);
          }
      FILE* f = fopen(file, "wb");
      if (f == nullptr)
        return 2;
      fwrite(table.data(), 1, table.size(), f);
      fclose(f);
      return 0;
    }
    else if (strcmp(argv[i], "--probe-continuous-trace") == 0) {
      // --probe-continuous-trace ox oy oz dx dy dz tmax material max_hits: after the scene is committed, Raytracing::continuous_trace
      // (rt.cxx:373-426) along that ray; prints PROBE_HITS {"count": n, "t": [...], "triangle": [...]} and exits (tests: which hits the reference
      // keeps when a probe ray of the Christensen-Burley gather meets more surfaces of its material than the buffer holds)
      for (int k = 0; k < 7; ++k)
        probe_ray[k] = float(atof(next()));
      probe_material = uint32_t(atoll(next()));
      probe_max_hits = uint32_t(atoll(next()));
    }
    else if (strcmp(argv[i], "--inject-density") == 0)
      inject_density = uint32_t(atoll(next()));
    else if (strcmp(argv[i], "--pixel-size") == 0)
      pixel_size = uint32_t(atoll(next()));
    else if (strcmp(argv[i], "--integrator") == 0)
      integrator_name = next();
    else if (strcmp(argv[i], "--spp") == 0)
      spp = atoll(next());
    else if (strcmp(argv[i], "--max-iterations") == 0)
      max_iterations = atoll(next());
    else if (strcmp(argv[i], "--adaptive-log") == 0)
      adaptive_log = next();
    else if (strcmp(argv[i], "--noise-threshold") == 0)  // Scene::noise_threshold (scene.hxx:45, default 0.1): 0 switches the adaptive sampling of CPUPathTracing off
      noise_threshold = float(atof(next()));
    else if (strcmp(argv[i], "--checkpoint") == 0)
      checkpoint_file = next();
    else if (strcmp(argv[i], "--resume") == 0)
      resume_file = next();
    else if (strcmp(argv[i], "--subsurface-class") == 0)
      subsurface_class = atoll(next());
    else if (strcmp(argv[i], "--out") == 0)
      out_file = next();
    else if (strcmp(argv[i], "--snapshot") == 0)
      snapshot_file = next();
    else if (strcmp(argv[i], "--data") == 0)
      data_folder = next();
    else if (strcmp(argv[i], "--opt") == 0) {
      std::string kv = next();
      auto eq = kv.find('=');
      if (eq != std::string::npos)
        opts.emplace_back(kv.substr(0, eq), kv.substr(eq + 1));
    } else {
      usage();
      return 1;
    }
  }
  if (scene_file.empty() && load_snapshot_file.empty()) {
    usage();
    return 1;
  }

  init_platform();
  env().setup(argv[0]);

  IORDatabase ior_database;
  Raytracing raytracing;
  SceneRepresentation scene(raytracing.scheduler(), ior_database);
  std::vector<uint8_t> snapshot_storage;
  const Scene* scene_ptr = &scene.scene();
  const Camera* camera_ptr = &scene.camera();

  if (load_snapshot_file.empty()) {
    ior_database.load((data_folder + "spectrum/").c_str());
    if (scene.load_from_file(scene_file.c_str(), SceneRepresentation::LoadEverything) == false) {
      printf("failed to load %s\n", scene_file.c_str());
      return 2;
    }
    if (spp > 0) {
      scene.mutable_scene().samples = uint32_t(spp);
    }
  } else {
    if (load_snapshot(load_snapshot_file.c_str(), snapshot_storage, scene_ptr, camera_ptr) == false) {
      printf("failed to load snapshot %s\n", load_snapshot_file.c_str());
      return 2;
    }
    if (spp > 0) {
      const_cast<Scene*>(scene_ptr)->samples = uint32_t(spp);
    }
  }
  if (inject_density > 0) {
    // The loader only reads .nvdb grids (medium_pool.cxx:93-99) and the tree ships none: give the scene's media the
    // state MediumPool::add leaves behind (medium_pool.cxx:41-58: density normalised to max 1, dimensions, class) with a
    // smooth procedural field, so that the heterogeneous branches of scene_medium.hxx run on both sides.
    const uint32_t n = inject_density;
    for (uint64_t mi = 0; mi < scene_ptr->mediums.count; ++mi) {
      Medium& m = const_cast<Medium&>(scene_ptr->mediums[mi]);
      float* grid = reinterpret_cast<float*>(malloc(sizeof(float) * n * n * n));
      float max_density = 0.0f;
      for (uint32_t z = 0; z < n; ++z)
        for (uint32_t y = 0; y < n; ++y)
          for (uint32_t x = 0; x < n; ++x) {
            float fx = (float(x) + 0.5f) / float(n), fy = (float(y) + 0.5f) / float(n), fz = (float(z) + 0.5f) / float(n);
            float blob0 = expf(-12.0f * (sqr(fx - 0.35f) + sqr(fy - 0.40f) + sqr(fz - 0.55f)));
            float blob1 = expf(-20.0f * (sqr(fx - 0.70f) + sqr(fy - 0.65f) + sqr(fz - 0.35f)));
            float waves = 0.15f * (1.0f + sinf(9.0f * fx) * sinf(7.0f * fy + 1.0f) * sinf(8.0f * fz + 2.0f));
            float d = blob0 + 0.8f * blob1 + waves;
            grid[x + y * n + z * n * n] = d;
            max_density = max(max_density, d);
          }
      for (uint32_t k = 0; k < n * n * n; ++k)
        grid[k] /= max_density;
      m.density.a = grid;
      m.density.count = uint64_t(n) * n * n;
      m.dimensions = {n, n, n};
      m.cls = Medium::Class::Heterogeneous;
    }
  }
  if (noise_threshold >= 0.0f)
    const_cast<Scene*>(scene_ptr)->noise_threshold = noise_threshold;
  if (subsurface_class >= 1) {  // the same geometry under the other subsurface class, without a second scene file (tests/test_gpu_sssmesh.py)
    auto& materials = const_cast<Scene*>(scene_ptr)->materials;
    for (uint64_t i = 0; i < materials.count; ++i) {
      if (materials[i].subsurface.cls != SubsurfaceMaterial::Class::Disabled)
        materials[i].subsurface.cls = SubsurfaceMaterial::Class(uint32_t(subsurface_class));
    }
  }
  raytracing.link_scene(*scene_ptr);
  raytracing.link_camera(*camera_ptr);
  raytracing.commit_changes();

  const Scene& sc = *scene_ptr;
  const Camera& cam = *camera_ptr;
  printf("scene: %llu vertices, %llu triangles, %llu materials, %llu emitters, %llu spectrums, %llu images, %llu mediums, radius %.4f, spectral %d\n",
    (unsigned long long)sc.vertices.count, (unsigned long long)sc.triangles.count, (unsigned long long)sc.materials.count, (unsigned long long)sc.emitter_instances.count,
    (unsigned long long)sc.spectrums.count, (unsigned long long)sc.images.count, (unsigned long long)sc.mediums.count, sc.bounding_sphere_radius, int(sc.spectral()));
  printf("film: %u x %u, samples %u, max path %u, rr start %u\n", cam.film_size.x, cam.film_size.y, sc.samples, sc.max_path_length, sc.random_path_termination);

  if (snapshot_file.empty() == false) {
    if (write_snapshot(snapshot_file.c_str(), sc, cam) == false) {
      printf("failed to write %s\n", snapshot_file.c_str());
      return 3;
    }
  }

  if (probe_max_hits > 0u) {
    std::vector<IntersectionBase> hits(probe_max_hits);
    const Ray ray = {{probe_ray[0], probe_ray[1], probe_ray[2]}, {probe_ray[3], probe_ray[4], probe_ray[5]}, kRayEpsilon, probe_ray[6]};
    ContinousTraceOptions ct = {hits.data(), probe_max_hits, probe_material};
    Sampler smp(1u, 2u);
    const uint32_t count = raytracing.continuous_trace(sc, ray, ct, smp);
    printf("PROBE_HITS {\"count\": %u, \"t\": [", count);
    for (uint32_t k = 0; k < count; ++k)
      printf("%s%.9g", k ? ", " : "", hits[k].t);
    printf("], \"triangle\": [");
    for (uint32_t k = 0; k < count; ++k)
      printf("%s%u", k ? ", " : "", hits[k].triangle_index);
    printf("]}\n");
    return 0;
  }

  if (integrator_name == "none")
    return 0;

  if ((integrator_name.rfind("hip-", 0) == 0) && (getenv("ETX_HIP_LIBRARY") == nullptr)) {
    // the in-tree library: <repo>/oracle/_ref/etx_oracle -> <repo>/etx-tracer_amd/libetx_hip.so
    char exe[4096] = {};
    const ssize_t len = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
    if (len > 0) {
      std::string dir(exe, size_t(len));
      dir = dir.substr(0, dir.find_last_of('/'));
      setenv("ETX_HIP_LIBRARY", (dir + "/../../etx-tracer_amd/libetx_hip.so").c_str(), 0);
    }
  }
#if !defined(ETX_DRIVER_HIP_ONLY)
  CPUPathTracing pt(raytracing);
  CPUVCM vcm(raytracing);
  CPUBidirectional bdpt(raytracing);
#endif
  HIPVCM hip_vcm(raytracing);          // the device integrators sit behind the same plugin interface (app.hxx:72-82)
  HIPPathTracing hip_pt(raytracing);
  HIPBidirectional hip_bdpt(raytracing);
  Integrator* integrator = nullptr;
#if !defined(ETX_DRIVER_HIP_ONLY)
  if (integrator_name == "pt")
    integrator = &pt;
  else if (integrator_name == "vcm")
    integrator = &vcm;
  else if (integrator_name == "bdpt")
    integrator = &bdpt;
  else
#endif
  if (integrator_name == "hip-vcm")
    integrator = &hip_vcm;
  else if (integrator_name == "hip-pt")
    integrator = &hip_pt;
  else if (integrator_name == "hip-bdpt")
    integrator = &hip_bdpt;
  else {
    usage();
    return 1;
  }

  for (const auto& kv : opts) {
    auto& o = integrator->options();
    // Options::make appends a SECOND entry when the class differs from the stored one and Options::get returns the first match
    // (options.hxx:121-128,141-152): "vcm-kernel" is stored as a bool (vcm_shared.cxx:43) but read as an integral (:19), so without
    // the removal --opt vcm-kernel=0 would be shadowed by the stored bool and the top-hat branch could not be reached
    o.remove(kv.first);
    if ((kv.second == "true") || (kv.second == "false"))
      o.set_bool(kv.first, kv.second == "true", kv.first);
    else if (kv.second.find('.') != std::string::npos)
      o.set_float(kv.first, float(atof(kv.second.c_str())), kv.first);
    else
      o.set_integral(kv.first, uint32_t(atoll(kv.second.c_str())), kv.first);
  }

  if (integrator->enabled() == false) {
    printf("integrator %s is not available on this machine\n", integrator->name());
    return 5;
  }
  if (pixel_size >= 1u)
    raytracing.film().set_pixel_size(pixel_size);  // takes effect at the next Film::clear (film.cxx:365)
  raytracing.film().clear(Film::ClearEverything);
  auto t0 = std::chrono::steady_clock::now();
  HIPIntegratorBase* hip = (integrator_name.rfind("hip-", 0) == 0) ? static_cast<HIPIntegratorBase*>(integrator) : nullptr;
  if ((hip == nullptr) && ((checkpoint_file.empty() == false) || (resume_file.empty() == false))) {
    printf("--checkpoint / --resume: the reference integrators have no checkpoint\n");
    return 1;
  }
  if (resume_file.empty() == false) {
    std::vector<uint8_t> blob;
    if (FILE* f = fopen(resume_file.c_str(), "rb")) {
      fseek(f, 0, SEEK_END);
      blob.resize(size_t(ftell(f)));
      fseek(f, 0, SEEK_SET);
      if (fread(blob.data(), 1, blob.size(), f) != blob.size())
        blob.clear();
      fclose(f);
    }
    if (blob.empty() || (hip->resume(blob) == false)) {
      printf("failed to resume from %s\n", resume_file.c_str());
      return 7;
    }
    printf("resumed at iteration %u\n", integrator->status().completed_iterations);
  } else {
    integrator->run();
  }
  if ((integrator->state() == Integrator::State::Stopped) && (integrator->status().completed_iterations == 0u)) {
    printf("integrator %s did not start (see the log above)\n", integrator->name());
    return 6;
  }
  // Adaptive sampling of CPUPathTracing (path_tracing.cxx:57-99, film.cxx:233-330): which pixels the next iteration samples is the film's
  // active mask; the driver reads it whenever the integrator reports another completed iteration (the mask changes after even iterations
  // from 32 on, an iteration of the test scenes takes milliseconds: the 200 us poll sees every one).
  std::vector<uint32_t> active_series;
  uint32_t logged_iterations = 0;
  auto log_active = [&]() {
    if (adaptive_log.empty())
      return;
    const uint32_t done = integrator->status().completed_iterations;
    if (done == logged_iterations)
      return;
    uint32_t active = 0;
    const uint32_t pixel_count = cam.film_size.x * cam.film_size.y;
    for (uint32_t i = 0; i < pixel_count; ++i) {
      uint2 location = {};
      active += raytracing.film().active_pixel(i, location) ? 1u : 0u;
    }
    for (; logged_iterations < done; ++logged_iterations)
      active_series.push_back(active);
  };
  while (integrator->state() != Integrator::State::Stopped) {
    integrator->update();
    log_active();
    if ((max_iterations > 0) && (int64_t(integrator->status().completed_iterations) >= max_iterations) && (integrator->state() == Integrator::State::Running)) {
      integrator->stop(Integrator::Stop::WaitForCompletion);
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  const auto& st = integrator->status();
  uint32_t threads = raytracing.scheduler().max_thread_count() - 2u;
  uint64_t pixels = uint64_t(cam.film_size.x) * cam.film_size.y;
  double msamples = double(pixels) * double(st.completed_iterations) / st.total_time / 1.0e6;
  printf("integrator %s: %u iterations, total_time %.4f s (wall %.4f s), %u threads, %.4f Msamples/s\n", integrator->name(), st.completed_iterations, st.total_time, wall, threads,
    msamples);
  uint64_t r0 = g_oracle_rays_trace.load(), r1 = g_oracle_rays_transmittance.load(), r2 = g_oracle_rays_material.load();
  if (r0 + r1 + r2 > 0) {
    double samples = double(pixels) * double(st.completed_iterations);
    printf("rays: trace %llu transmittance %llu material %llu -> %.3f rays/sample\n", (unsigned long long)r0, (unsigned long long)r1, (unsigned long long)r2,
      double(r0 + r1 + r2) / samples);
  }
  if (adaptive_log.empty() == false) {
    log_active();
    // row k: pixels still active once k + 1 iterations were complete = what iteration k + 1 samples; iteration 0 samples every pixel
    if (FILE* f = fopen(adaptive_log.c_str(), "w")) {
      unsigned long long total = pixels;
      for (size_t k = 0; k + 1u < active_series.size(); ++k)
        total += active_series[k];
      fprintf(f, "# completed_iterations active_pixels_afterwards ; sampled pixel-iterations of the render: %llu\n", total);
      for (size_t k = 0; k < active_series.size(); ++k)
        fprintf(f, "%zu %u\n", k + 1u, active_series[k]);
      fclose(f);
    }
  }
  printf("adaptive sampling: %u of %u pixels converged at the last noise estimate (Film::active_pixel_count)\n", raytracing.film().active_pixel_count(), uint32_t(pixels));
  // machine readable line for bench.py / tests
  printf("ORACLE_RESULT {\"integrator\": \"%s\", \"iterations\": %u, \"seconds\": %.6f, \"threads\": %u, \"msamples_per_s\": %.6f, \"width\": %u, \"height\": %u}\n",
    integrator_name.c_str(), st.completed_iterations, st.total_time, threads, msamples, cam.film_size.x, cam.film_size.y);

  if (checkpoint_file.empty() == false) {
    std::vector<uint8_t> blob;
    FILE* f = hip->save_checkpoint(blob) ? fopen(checkpoint_file.c_str(), "wb") : nullptr;
    const bool written = (f != nullptr) && (fwrite(blob.data(), 1, blob.size(), f) == blob.size());
    if (f != nullptr)
      fclose(f);
    if (written == false) {
      printf("failed to write %s\n", checkpoint_file.c_str());
      return 4;
    }
  }
  if (out_file.empty() == false) {
    if (write_film(out_file.c_str(), raytracing.film(), st.completed_iterations, st.total_time, threads) == false) {
      printf("failed to write %s\n", out_file.c_str());
      return 4;
    }
  }
  return 0;
}
