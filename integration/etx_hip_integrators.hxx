// etx_hip_integrators.hxx - the reference-side binding of libetx_hip.so: the file a maintainer adds to etx-tracer as
// sources/etx/rt/integrators/hip_integrators.hxx. It is COMPILED here: oracle/build_ref.sh builds it
// against the reference's own headers into the headless driver (oracle/_ref/etx_oracle --integrator hip-vcm | hip-pt | hip-bdpt), and
// tests/test_gpu_binding.py renders through it on the GPU box and compares with the ctypes binding.
//
// Three classes on the reference's plugin interface `struct Integrator` (sources/etx/rt/integrators/integrator.hxx:12-98):
//   HIPVCM          in place of CPUVCM          (sources/etx/rt/integrators/vcm_cpu.cxx:243-310)
//   HIPPathTracing  in place of CPUPathTracing  (sources/etx/rt/integrators/path_tracing.cxx:112-172)
//   HIPBidirectional in place of CPUBidirectional (sources/etx/rt/integrators/bidirectional.cxx:1490-1560)
// Only API that exists in the reference is used. The film is published through Film::accumulate_camera_image /
// atomic_add_light_iteration / commit_light_iteration (film.hxx:57-59): Film::clear resets the per-pixel sample counts,
// so ONE accumulate per pixel stores the device's running mean verbatim (film.cxx:195-199), and one light "iteration"
// committed with index 0 stores the device's light image (film.cxx:332-343). That is O(pixels) host work per publish -
// off the hot path, done every kPublishInterval iterations and at the end; integration/film_merge_iteration.patch is
// the additive bulk interface that removes it.
//
// The library is loaded with dlopen (ETX_HIP_LIBRARY or next to the executable), so a host built with this file still
// starts on a machine without the backend: enabled() is false there and run() stays Stopped (integrator.hxx:49-51).
#pragma once

#include <etx/core/core.hxx>
#include <etx/render/host/film.hxx>
#include <etx/render/shared/spectrum.hxx>
#include <etx/rt/integrators/integrator.hxx>
#include <etx/rt/shared/path_tracing_shared.hxx>
#include <etx/rt/shared/vcm_shared.hxx>
#include <bluenoise.hxx>

#include <etx_hip.h>

#include <dlfcn.h>

#include <string>
#include <vector>

namespace etx {

// log::output walks its va_list twice (core/log.cxx:19-24), which only works where va_list is a plain pointer: texts go
// through as the format string itself, with '%' escaped.
inline void hip_report_error(const char* text) {
  std::string escaped;
  for (const char* c = (text != nullptr) ? text : "unknown error"; *c != 0; ++c) {
    escaped += *c;
    if (*c == '%')
      escaped += '%';
  }
  log::error(escaped.c_str());
}

// Entry points of include/etx_hip.h, resolved at run time.
struct HIPBackendLibrary {
  void* handle = nullptr;
  decltype(&etx_hip_create) create = nullptr;
  decltype(&etx_hip_destroy) destroy = nullptr;
  decltype(&etx_hip_last_error) last_error = nullptr;
  decltype(&etx_hip_upload_scene) upload_scene = nullptr;
  decltype(&etx_hip_update_scene) update_scene = nullptr;
  decltype(&etx_hip_set_bvh_builder) set_bvh_builder = nullptr;
  decltype(&etx_hip_upload_bluenoise) upload_bluenoise = nullptr;
  decltype(&etx_hip_upload_cie_table) upload_cie_table = nullptr;
  decltype(&etx_hip_upload_rgb_response) upload_rgb_response = nullptr;
  decltype(&etx_hip_begin) begin = nullptr;
  decltype(&etx_hip_try_render_iteration) try_render_iteration = nullptr;
  decltype(&etx_hip_poll) poll = nullptr;
  decltype(&etx_hip_sync) sync = nullptr;
  decltype(&etx_hip_read_film) read_film = nullptr;
  decltype(&etx_hip_read_film_begin) read_film_begin = nullptr;
  decltype(&etx_hip_read_film_end) read_film_end = nullptr;
  decltype(&etx_hip_stats) stats = nullptr;
  decltype(&etx_hip_checkpoint_bytes) checkpoint_bytes = nullptr;
  decltype(&etx_hip_checkpoint_save) checkpoint_save = nullptr;
  decltype(&etx_hip_checkpoint_load) checkpoint_load = nullptr;

  static HIPBackendLibrary& get() {
    static HIPBackendLibrary lib = load();
    return lib;
  }

  bool ok() const {
    return handle != nullptr;
  }

 private:
  static HIPBackendLibrary load() {
    HIPBackendLibrary lib;
    const char* path = getenv("ETX_HIP_LIBRARY");
    lib.handle = dlopen(path ? path : "libetx_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (lib.handle == nullptr) {
      hip_report_error((std::string("HIP backend not available: ") + dlerror()).c_str());
      return lib;
    }
    bool complete = true;
    auto resolve = [&](auto& fn, const char* name) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib.handle, name));
      complete = complete && (fn != nullptr);
    };
    resolve(lib.create, "etx_hip_create");
    resolve(lib.destroy, "etx_hip_destroy");
    resolve(lib.last_error, "etx_hip_last_error");
    resolve(lib.upload_scene, "etx_hip_upload_scene");
    resolve(lib.update_scene, "etx_hip_update_scene");
    resolve(lib.set_bvh_builder, "etx_hip_set_bvh_builder");
    resolve(lib.upload_bluenoise, "etx_hip_upload_bluenoise");
    resolve(lib.upload_cie_table, "etx_hip_upload_cie_table");
    resolve(lib.upload_rgb_response, "etx_hip_upload_rgb_response");
    resolve(lib.begin, "etx_hip_begin");
    resolve(lib.try_render_iteration, "etx_hip_try_render_iteration");
    resolve(lib.poll, "etx_hip_poll");
    resolve(lib.sync, "etx_hip_sync");
    resolve(lib.read_film, "etx_hip_read_film");
    resolve(lib.read_film_begin, "etx_hip_read_film_begin");
    resolve(lib.read_film_end, "etx_hip_read_film_end");
    resolve(lib.stats, "etx_hip_stats");
    resolve(lib.checkpoint_bytes, "etx_hip_checkpoint_bytes");
    resolve(lib.checkpoint_save, "etx_hip_checkpoint_save");
    resolve(lib.checkpoint_load, "etx_hip_checkpoint_load");
    if (complete == false) {
      log::error("libetx_hip.so misses entry points of etx_hip.h");
      dlclose(lib.handle);
      lib.handle = nullptr;
    }
    return lib;
  }
};

static_assert(sizeof(Scene) == sizeof(etx_abi_scene), "etx::Scene is the ABI struct");
static_assert(sizeof(Camera) == sizeof(etx_abi_camera), "etx::Camera is the ABI struct");
static_assert(sizeof(VCMOptions) == sizeof(etx_abi_vcm_options), "VCMOptions is passed by value");
static_assert(sizeof(PTOptions) == sizeof(etx_abi_pt_options), "PTOptions is passed by value");

struct HIPIntegratorBase : public Integrator {
  static constexpr uint32_t kPublishInterval = 16;  // iterations between two film publishes while rendering

  HIPIntegratorBase(Raytracing& r)
    : Integrator(r) {
  }

  ~HIPIntegratorBase() override {
    if (ctx != nullptr)
      HIPBackendLibrary::get().destroy(ctx);
  }

  bool enabled() const override {
    return HIPBackendLibrary::get().ok();
  }

  const Status& status() const override {
    return _status;
  }

  // CPUVCM::run / CPUPathTracing::run (vcm_cpu.cxx:255-262, path_tracing.cxx:128-135)
  void run() override {
    stop(Stop::Immediate);
    auto& lib = HIPBackendLibrary::get();
    if ((lib.ok() == false) || (can_run() == false))
      return;
    if ((ctx == nullptr) && (lib.create(0, &ctx) != ETX_HIP_OK)) {
      hip_report_error(lib.last_error(nullptr));
      ctx = nullptr;
      return;
    }
    // etx::Scene / etx::Camera ARE the ABI structs: the backend borrows them during the call and owns device copies
    // afterwards (deep copy + BVH build: Raytracing::commit_changes, rt.cxx:58-88, and the disabled sketch rt.cxx:141-238)
    // A host that knows what it edited since the last run (scene_edited) keeps geometry, BVH and images on the device and has moved
    // vertices refit there; otherwise everything is uploaded, which is what commit_changes does on every change (app.cxx:368-399)
    const auto* scene_abi = reinterpret_cast<const etx_abi_scene*>(&rt.scene());
    const auto* camera_abi = reinterpret_cast<const etx_abi_camera*>(&rt.camera());
    // option "hip-device_bvh": the tree is built on the device (linear BVH: a millisecond for a million triangles, traverses at
    // ~0.7 of the rate of the host's binned-SAH tree) - for geometry that changes, or for time to first image
    lib.set_bvh_builder(ctx, integrator_options.get_bool("hip-device_bvh", false) ? ETX_HIP_BVH_DEVICE_LBVH : ETX_HIP_BVH_HOST_SAH);
    const bool in_place = scene_on_device && (pending_changes != kEverythingChanged);
    const int uploaded = in_place ? lib.update_scene(ctx, scene_abi, camera_abi, pending_changes) : lib.upload_scene(ctx, scene_abi, camera_abi);
    pending_changes = kEverythingChanged;
    scene_on_device = uploaded == ETX_HIP_OK;
    if (uploaded != ETX_HIP_OK) {
      hip_report_error(lib.last_error(ctx));  // e.g. ETX_HIP_ERROR_UNSUPPORTED: stay Stopped, like a failed Embree commit
      return;
    }
    if (rt.scene().spectral() && ((upload_cie_table() == false) || (upload_rgb_response() == false)))
      return;
    if (begin() == false)
      return;
    rt.film().clear(Film::ClearCameraData | Film::ClearLightData);
    _status = {};
    submitted = 0;
    published = 0;
    progressive_stage = 0;
    current_state = State::Running;
  }

  // The host edited the committed scene in place since the last run(): ETX_HIP_CHANGED_CAMERA | _MATERIALS | _POSITIONS (include/etx_hip.h).
  // The next run() then calls etx_hip_update_scene instead of etx_hip_upload_scene. Anything else (topology, images) needs no call:
  // without one, run() uploads the whole scene.
  void scene_edited(uint32_t changed) {
    pending_changes = (pending_changes == kEverythingChanged) ? changed : (pending_changes | changed);
  }

  // Checkpoint / resume. The reference has neither (a stopped render starts over, app.cxx:193-216): the film state
  // of the render in progress - sums, per-pixel sample counts, adaptive-sampling state, next iteration - as one buffer the caller
  // stores. Waits for the iterations in flight; rendering may go on afterwards.
  bool save_checkpoint(std::vector<uint8_t>& blob) {
    auto& lib = HIPBackendLibrary::get();
    if (ctx == nullptr)
      return false;
    blob.resize(lib.checkpoint_bytes(ctx));
    if (blob.empty() || (lib.checkpoint_save(ctx, blob.data(), blob.size()) != ETX_HIP_OK)) {
      hip_report_error(lib.last_error(ctx));
      return false;
    }
    return true;
  }

  // run() continued from a checkpoint of the same scene, options and film size: the completed iterations are not rendered again,
  // the remaining ones carry their own indices (and with them their seeds, radii and MIS weights)
  bool resume(const std::vector<uint8_t>& blob) {
    run();
    if (current_state != State::Running)
      return false;
    auto& lib = HIPBackendLibrary::get();
    if (lib.checkpoint_load(ctx, blob.data(), blob.size()) != ETX_HIP_OK) {
      hip_report_error(lib.last_error(ctx));
      current_state = State::Stopped;
      return false;
    }
    read_status();
    submitted = _status.completed_iterations;
    publish_film();
    return true;
  }

  // CPUVCM::update (vcm_cpu.cxx:264-276): called once per GUI frame, must not block
  void update() override {
    if (current_state == State::Stopped)
      return;
    auto& lib = HIPBackendLibrary::get();
    const uint32_t total = rt.scene().samples;
    if ((current_state == State::Running) && (submitted < total)) {
      const int rc = lib.try_render_iteration(ctx);  // 1: handed to a free device lane, 0: every lane is busy
      if (rc < 0) {
        hip_report_error(lib.last_error(ctx));
        current_state = State::Stopped;
        return;
      }
      submitted += uint32_t(rc);
    }
    read_status();
    const bool all_submitted = (submitted >= total) || (current_state == State::WaitingForCompletion);
    const int idle = lib.poll(ctx);  // 1 = every iteration handed over so far has finished
    if (idle < 0) {
      hip_report_error(lib.last_error(ctx));
      current_state = State::Stopped;
      return;
    }
    if (all_submitted && (idle == 1)) {
      read_status();
      publish_film();
      current_state = State::Stopped;  // vcm_cpu.cxx:234
    } else {
      progressive_publish();
    }
  }

  // While rendering: the camera and the light layer are fetched one after the other with the asynchronous read-back
  // (etx_hip_read_film_begin / _end with wait = 0: never blocks the caller's frame), then written to the Film.
  void progressive_publish() {
    auto& lib = HIPBackendLibrary::get();
    const size_t pixels = size_t(rt.film().size().x) * rt.film().size().y;
    if (progressive_stage == 0) {
      if (_status.completed_iterations < published + kPublishInterval)
        return;
      camera.resize(pixels);
      light.resize(pixels);
      if (lib.read_film_begin(ctx, ETX_HIP_LAYER_CAMERA) == ETX_HIP_OK)
        progressive_stage = 1;
      progressive_iterations = _status.completed_iterations;
    } else if (progressive_stage == 1) {
      if (lib.read_film_end(ctx, &camera[0].x, pixels * sizeof(float4), 0) == 1)
        progressive_stage = (writes_light_image() && (lib.read_film_begin(ctx, ETX_HIP_LAYER_LIGHT) == ETX_HIP_OK)) ? 2 : 3;
    } else if (progressive_stage == 2) {
      if (lib.read_film_end(ctx, &light[0].x, pixels * sizeof(float4), 0) == 1)
        progressive_stage = 3;
    }
    if (progressive_stage == 3) {
      write_layers_to_film(writes_light_image(), false);
      published = progressive_iterations;
      progressive_stage = 0;
    }
  }

  // vcm_cpu.cxx:278-288
  void stop(Stop st) override {
    if (current_state == State::Stopped)
      return;
    if (st == Stop::Immediate) {
      if (ctx != nullptr) {
        HIPBackendLibrary::get().sync(ctx);
        read_status();
        publish_film();
      }
      current_state = State::Stopped;
    } else {
      current_state = State::WaitingForCompletion;
    }
  }

  void update_options() override {
    if (current_state == State::Running)
      run();
  }

  bool have_updated_camera_image() const override {
    const bool r = camera_updated;
    camera_updated = false;
    return r;
  }

  bool have_updated_light_image() const override {
    const bool r = light_updated;
    light_updated = false;
    return r;
  }

 protected:
  virtual bool begin() = 0;
  virtual bool writes_light_image() const = 0;
  virtual bool converges() const {  // the integrator honours Scene::noise_threshold (only CPUPathTracing estimates noise levels)
    return false;
  }

  void read_status() {
    etx_hip_stats_t s = {};
    if (HIPBackendLibrary::get().stats(ctx, &s, sizeof(s)) != ETX_HIP_OK)
      return;
    _status.last_iteration_time = s.last_iteration_time;  // Integrator::Status, integrator.hxx:24-37
    _status.total_time = s.total_time;
    _status.completed_iterations = s.completed_iterations;
    _status.current_iteration = s.current_iteration;
    // path tracing with adaptive sampling: an iteration that sampled no pixel ends the render (path_tracing.cxx:91-93)
    if ((rt.scene().noise_threshold > 0.0f) && (s.completed_iterations > 33u) && (s.last_active_pixels == 0u) && converges() && (current_state == State::Running))
      current_state = State::WaitingForCompletion;
  }

  // The device keeps the running means; mirror them into the Film through its per-pixel interface.
  void publish_film() {
    auto& lib = HIPBackendLibrary::get();
    Film& film = rt.film();
    const uint2 dim = film.size();
    const size_t pixels = size_t(dim.x) * dim.y;
    if (_status.completed_iterations == 0)
      return;
    // Film::pixel_size() > 1 (the GUI sets 8 while the camera moves, app.cxx:135): the reference renders one path per BLOCK and
    // accumulate_camera_image / atomic_add_light_iteration fill the block (film.cxx:152-168, 185-199). The device renders the full
    // frame in either case - an iteration costs it milliseconds -, so the preview is published as block means through those same
    // two calls (write_layers_to_film); the bulk interface below writes single pixels and is used at pixel size 1 only.
    const bool preview = film.pixel_size() != 1u;
    camera.resize(pixels);
    normal.resize(pixels);
    albedo.resize(pixels);
    if (lib.read_film(ctx, ETX_HIP_LAYER_CAMERA, &camera[0].x, pixels * sizeof(float4)) != ETX_HIP_OK) {
      hip_report_error(lib.last_error(ctx));
      return;
    }
    const bool aovs = writes_light_image() == false;  // the path tracer fills the denoiser AOVs (film.cxx:207-216)
    if (aovs && ((lib.read_film(ctx, ETX_HIP_LAYER_NORMAL, &normal[0].x, pixels * sizeof(float4)) != ETX_HIP_OK) ||
                 (lib.read_film(ctx, ETX_HIP_LAYER_ALBEDO, &albedo[0].x, pixels * sizeof(float4)) != ETX_HIP_OK))) {
      hip_report_error(lib.last_error(ctx));
      return;
    }
#if defined(ETX_FILM_HAS_MERGE_ITERATION)
    // integration/film_merge_iteration.patch applied: one bulk call, no per-pixel accumulate
    if (preview == false) {
    if (writes_light_image()) {
      light.resize(pixels);
      if (lib.read_film(ctx, ETX_HIP_LAYER_LIGHT, &light[0].x, pixels * sizeof(float4)) != ETX_HIP_OK) {
        hip_report_error(lib.last_error(ctx));
        return;
      }
    }
    if (aovs) {
      for (size_t i = 0; i < pixels; ++i)  // Film::layer(Normals) = n * 0.5 + 0.5: back to the stored normal
        normal[i] = {normal[i].x * 2.0f - 1.0f, normal[i].y * 2.0f - 1.0f, normal[i].z * 2.0f - 1.0f, 0.0f};
    }
    film.merge_iteration(camera.data(), writes_light_image() ? light.data() : nullptr, aovs ? normal.data() : nullptr, aovs ? albedo.data() : nullptr, _status.completed_iterations);
    camera_updated = true;
    light_updated = writes_light_image();
    published = _status.completed_iterations;
    return;
    }
#endif
    (void)preview;
    if (writes_light_image()) {
      light.resize(pixels);
      if (lib.read_film(ctx, ETX_HIP_LAYER_LIGHT, &light[0].x, pixels * sizeof(float4)) != ETX_HIP_OK) {
        hip_report_error(lib.last_error(ctx));
        return;
      }
    }
    write_layers_to_film(writes_light_image(), aovs);
    published = _status.completed_iterations;
  }

  // camera (+ light, + normal / albedo) buffers -> Film, through the per-pixel interface the reference has
  void write_layers_to_film(bool with_light, bool aovs) {
    Film& film = rt.film();
    const uint2 dim = film.size();
    film.clear(Film::ClearCameraData | (with_light ? uint32_t(Film::ClearLightData) : 0u));
    // Film::clear applies a pending set_pixel_size (film.cxx:365): read it afterwards. One call per block of pixel_size^2 pixels
    // (= per pixel at size 1) with the mean of the device's full-resolution film over the block; both Film calls fill the block.
    const uint32_t block = film.pixel_size();
    // mean of a layer over the block whose base pixel is (bx, by); rows of the device film are the Film's storage rows: storage
    // row r holds pixel y = H - 1 - r (film.cxx:189)
    auto block_mean = [&](const std::vector<float4>& layer, uint32_t bx, uint32_t by) {
      float3 sum = {};
      uint32_t n = 0;
      for (uint32_t y = by, ye = min(by + block, dim.y); y < ye; ++y) {
        for (uint32_t x = bx, xe = min(bx + block, dim.x); x < xe; ++x, ++n) {
          const float4& v = layer[size_t(dim.y - 1u - y) * dim.x + x];
          sum += float3{v.x, v.y, v.z};
        }
      }
      return sum / float(max(n, 1u));
    };
    for (uint32_t by = 0; by < dim.y; by += block) {
      for (uint32_t bx = 0; bx < dim.x; bx += block) {
        const float3 c = block_mean(camera, bx, by);
        // Film::layer(Normals) = n * 0.5 + 0.5 (film.cxx:411): the device returns the layer, the film stores n
        const float3 n = aovs ? block_mean(normal, bx, by) * 2.0f - float3{1.0f, 1.0f, 1.0f} : float3{};
        const float3 a = aovs ? block_mean(albedo, bx, by) : float3{};
        film.accumulate_camera_image({bx, by}, c, n, a);
      }
    }
    camera_updated = true;
    if (with_light) {
      // atomic_add_light_iteration maps ndc -> reduced-resolution pixel -> block (film.cxx:152-159): aim at the block's centre
      const uint2 reduced = film.dimensions();
      for (uint32_t by = 0; by < dim.y; by += block) {
        for (uint32_t bx = 0; bx < dim.x; bx += block) {
          const float3 l = block_mean(light, bx, by);
          const float2 ndc = {(float(bx / block) + 0.5f) / float(reduced.x) * 2.0f - 1.0f, (float(by / block) + 0.5f) / float(reduced.y) * 2.0f - 1.0f};
          film.atomic_add_light_iteration(l, ndc);
        }
      }
      film.commit_light_iteration(0);
      light_updated = true;
    }
  }

  // sample_blue_noise (path_tracing.cxx:173-178) is a function of (pixel & 127, sample & 255, dimension & 7) only and its
  // tables are private to thirdparty/bluenoise: tabulate the sampler for the class BNSampler picks for scene.samples
  bool upload_bluenoise() {
    const uint32_t samples = min(max(rt.scene().samples, 1u), 256u);
    uint32_t set = 0;
    while ((1u << set) < samples)
      ++set;
    if (uploaded_bluenoise_sets & (1u << set))
      return true;
    std::vector<uint8_t> values(size_t(128) * 128 * 256 * 8);
    for (uint32_t py = 0; py < 128; ++py)
      for (uint32_t px = 0; px < 128; ++px)
        for (uint32_t sample = 0; sample < 256; ++sample) {
          BNSampler smp(px, py, rt.scene().samples, sample);
          for (uint32_t d = 0; d < 8; ++d)
            values[(((size_t(py) * 128 + px) * 256) + sample) * 8 + d] = uint8_t(smp.get(d) * 256.0f);
        }
    if (HIPBackendLibrary::get().upload_bluenoise(ctx, set, values.data(), values.size()) != ETX_HIP_OK) {
      hip_report_error(HIPBackendLibrary::get().last_error(ctx));
      return false;
    }
    uploaded_bluenoise_sets |= 1u << set;
    return true;
  }

  // the CIE observer behind SpectralResponse::to_rgb (spectrum.hxx:28-140, 271-293): the host's data, not compiled into the backend
  // apply_rgb (scene.hxx:249-260): rgb_response of the unit colours at every integer wavelength = the rows of its table
  bool upload_rgb_response() {
    std::vector<float> rows(size_t(spectrum::RGBResponseWavelengthCount) * 3u);
    for (uint32_t k = 0; k < spectrum::RGBResponseWavelengthCount; ++k) {
      const SpectralQuery q = {spectrum::kRGBResponseShortestWavelength + float(k), SpectralQuery::Spectral};
      rows[3u * k + 0u] = rgb_response(q, {1.0f, 0.0f, 0.0f}).value;
      rows[3u * k + 1u] = rgb_response(q, {0.0f, 1.0f, 0.0f}).value;
      rows[3u * k + 2u] = rgb_response(q, {0.0f, 0.0f, 1.0f}).value;
    }
    if (HIPBackendLibrary::get().upload_rgb_response(ctx, rows.data(), spectrum::RGBResponseWavelengthCount, spectrum::kRGBResponseShortestWavelength) != ETX_HIP_OK) {
      hip_report_error(HIPBackendLibrary::get().last_error(ctx));
      return false;
    }
    return true;
  }

  bool upload_cie_table() {
    std::vector<float> xyz(size_t(spectrum::WavelengthCount) * 3u);
    for (uint32_t k = 0; k < spectrum::WavelengthCount; ++k) {
      const float3 v = spectrum::spectral_xyz(k);
      xyz[3u * k + 0u] = v.x, xyz[3u * k + 1u] = v.y, xyz[3u * k + 2u] = v.z;
    }
    if (HIPBackendLibrary::get().upload_cie_table(ctx, xyz.data(), spectrum::WavelengthCount, spectrum::kShortestWavelength) != ETX_HIP_OK) {
      hip_report_error(HIPBackendLibrary::get().last_error(ctx));
      return false;
    }
    return true;
  }

  etx_hip_context* ctx = nullptr;
  static constexpr uint32_t kEverythingChanged = 0xffffffffu;
  // Options of the backend itself, next to the reference integrator's own keys (they show up in the GUI's option panel like any other).
  // "hip-reference_seeding": the camera path of pixel i starts from the sampler state of light path i, as CPUVCM / CPUBidirectional do
  // (vcm_shared.hxx:312,357; bidirectional.cxx:377-378). Off by default: the device's ray queries draw nothing for opaque triangles, so with
  // shared seeds the two paths of a pixel stay aligned draw for draw and their vertex connections are correlated; on, the
  // device renders the unmodified reference's estimator (asserted against the pinned reference, tests/test_gpu_options.py).
  static constexpr const char* kReferenceSeedingKey = "hip-reference_seeding";
  void add_backend_options() {
    integrator_options.set_bool(kReferenceSeedingKey, false, "Reference seeding (light / camera paths share their stream)");
  }

  uint32_t pending_changes = kEverythingChanged;
  bool scene_on_device = false;
  Status _status = {};
  uint32_t submitted = 0, published = 0;
  uint32_t progressive_stage = 0, progressive_iterations = 0;  // asynchronous publish while rendering (progressive_publish)
  uint32_t uploaded_bluenoise_sets = 0;
  std::vector<float4> camera, light, normal, albedo;
  mutable bool camera_updated = false, light_updated = false;
};

struct HIPVCM : public HIPIntegratorBase {
  HIPVCM(Raytracing& r)
    : HIPIntegratorBase(r) {
    VCMOptions::default_values().store(integrator_options);  // the option keys of CPUVCM (vcm_shared.cxx:30-47)
    add_backend_options();
  }

  const char* name() override {
    return "VCM (HIP gfx950)";
  }

  const char* status_str() const override {
    return (current_state == State::Stopped) ? "Stopped" : "Rendering on the device";
  }

 protected:
  bool begin() override {
    VCMOptions loaded = VCMOptions::default_values();
    loaded.load(integrator_options);
    if (loaded.blue_noise && (upload_bluenoise() == false))
      return false;
    // etx_abi_vcm_options IS VCMOptions plus one byte in its tail padding (static_assert above, oracle/ref/abi_check.cxx): the members are
    // copied one by one, so whatever the padding of `loaded` holds never reaches the backend
    etx_abi_vcm_options opt = {};
    opt.options = loaded.options, opt.radius_decay = loaded.radius_decay, opt.kernel = loaded.kernel;
    opt.initial_radius = loaded.initial_radius, opt.blue_noise = loaded.blue_noise ? 1u : 0u;
    opt.reference_seeding = integrator_options.get_bool(kReferenceSeedingKey, false) ? 1u : 0u;
    if (HIPBackendLibrary::get().begin(ctx, ETX_HIP_INTEGRATOR_VCM, &opt, sizeof(opt), /* first iteration */ 0, /* stride */ 1) != ETX_HIP_OK) {
      hip_report_error(HIPBackendLibrary::get().last_error(ctx));
      return false;
    }
    return true;
  }

  bool writes_light_image() const override {
    return true;
  }
};

struct HIPPathTracing : public HIPIntegratorBase {
  HIPPathTracing(Raytracing& r)
    : HIPIntegratorBase(r) {
    // the option keys of CPUPathTracing (path_tracing.cxx:137-142)
    integrator_options.set_bool("direct", true, "Direct Hits");
    integrator_options.set_bool("nee", true, "Light Sampling");
    integrator_options.set_bool("mis", true, "Multiple Importance Sampling");
    integrator_options.set_bool("bn", true, "Use Blue Noise");
  }

  const char* name() override {
    return "Path Tracing (HIP gfx950)";
  }

  const char* status_str() const override {
    return (current_state == State::Stopped) ? "Stopped" : "Rendering on the device";
  }

 protected:
  bool begin() override {
    PTOptions opt = {};
    opt.direct = integrator_options.get_bool("direct", opt.direct);  // CPUPathTracingImpl::start, path_tracing.cxx:36-40
    opt.nee = integrator_options.get_bool("nee", opt.nee);
    opt.mis = integrator_options.get_bool("mis", opt.mis);
    opt.blue_noise = integrator_options.get_bool("bn", opt.blue_noise);
    if (opt.blue_noise && (upload_bluenoise() == false))
      return false;
    if (HIPBackendLibrary::get().begin(ctx, ETX_HIP_INTEGRATOR_PT, &opt, sizeof(opt), 0, 1) != ETX_HIP_OK) {
      hip_report_error(HIPBackendLibrary::get().last_error(ctx));
      return false;
    }
    return true;
  }

  bool writes_light_image() const override {
    return false;
  }

  bool converges() const override {
    return true;
  }
};

struct HIPBidirectional : public HIPIntegratorBase {
  HIPBidirectional(Raytracing& r)
    : HIPIntegratorBase(r) {
    // the option keys and defaults of CPUBidirectional (bidirectional.cxx:1443-1466, mode BDPTFast :332)
    integrator_options.set_integral("bdpt-mode", uint32_t(ETX_BDPT_MODE_FAST), "Mode");
    integrator_options.set_bool("bdpt-conn_direct_hit", true, "Direct Hits");
    integrator_options.set_bool("bdpt-conn_connect_to_camera", true, "Light Path to Camera");
    integrator_options.set_bool("bdpt-conn_connect_to_light", true, "Camera Path to Light");
    integrator_options.set_bool("bdpt-conn_connect_vertices", true, "Camera Path to Light Path");
    integrator_options.set_bool("bdpt-conn_mis", true, "Multiple Importance Sampling");
    integrator_options.set_bool("bdpt-blue_noise", true, "Enable Blue Noise");
    add_backend_options();
  }

  const char* name() override {
    return "Bidirectional (HIP gfx950)";
  }

  const char* status_str() const override {
    return (current_state == State::Stopped) ? "Stopped" : "Rendering on the device";
  }

 protected:
  bool begin() override {
    etx_abi_bdpt_options opt = {};
    opt.mode = integrator_options.get_integral("bdpt-mode", uint32_t(ETX_BDPT_MODE_FAST));  // CPUBidirectionalImpl::start, :1469-1478
    opt.direct_hit = integrator_options.get_bool("bdpt-conn_direct_hit", true);
    opt.connect_to_camera = integrator_options.get_bool("bdpt-conn_connect_to_camera", true);
    opt.connect_to_light = integrator_options.get_bool("bdpt-conn_connect_to_light", true);
    opt.connect_vertices = integrator_options.get_bool("bdpt-conn_connect_vertices", true);
    opt.mis = integrator_options.get_bool("bdpt-conn_mis", true);
    opt.blue_noise = integrator_options.get_bool("bdpt-blue_noise", true);
    opt.reference_seeding = integrator_options.get_bool(kReferenceSeedingKey, false) ? 1u : 0u;
    if (opt.blue_noise && (upload_bluenoise() == false))
      return false;
    if (HIPBackendLibrary::get().begin(ctx, ETX_HIP_INTEGRATOR_BDPT, &opt, sizeof(opt), 0, 1) != ETX_HIP_OK) {
      hip_report_error(HIPBackendLibrary::get().last_error(ctx));
      return false;
    }
    return true;
  }

  bool writes_light_image() const override {
    return true;
  }
};

}  // namespace etx
