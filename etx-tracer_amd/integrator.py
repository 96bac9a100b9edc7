"""Host-side mirror of the reference's integrator plugin interface for the HIP backend.

`Integrator` restates `struct Integrator` (sources/etx/rt/integrators/integrator.hxx:12-98): name / run / update /
stop / status / options / state, same method names and call protocol as the C++ class a maintainer adds to the
raytracer (INTEGRATION.md shows that class; both sit on the same C ABI). `HIPVCM` takes the place of `CPUVCM`
(sources/etx/rt/integrators/vcm_cpu.cxx:243-310): run() arms the device pipeline, update() renders / polls one
iteration per call and stops at scene.samples, stop() drains the device.
"""
import enum
import time

from . import api


class State(enum.Enum):  # Integrator::State, integrator.hxx:13-17
    Stopped = 0
    Running = 1
    WaitingForCompletion = 2


class Stop(enum.Enum):  # Integrator::Stop, integrator.hxx:19-22
    Immediate = 0
    WaitForCompletion = 1


class Integrator:
    def __init__(self):
        self.current_state = State.Stopped
        self.integrator_options = {}

    def name(self):
        return "Basic Integrator"

    def enabled(self):
        return True

    def options(self):
        return self.integrator_options

    def state(self):
        return self.current_state

    def run(self):
        pass

    def update(self):
        pass

    def stop(self, how=Stop.Immediate):
        pass

    def update_options(self):
        if self.current_state == State.Running:
            self.run()


def vcm_options_from_dict(values):
    """VCMOptions::load (sources/etx/rt/integrators/vcm_shared.cxx:15-31): same keys, same defaults."""
    o = api.VCMOptions.default_values()
    o.initial_radius = float(values.get("vcm-initial_radius", o.initial_radius))
    o.radius_decay = int(values.get("vcm-radius_decay", o.radius_decay))
    o.blue_noise = 1 if values.get("vcm-blue_noise", bool(o.blue_noise)) else 0
    o.kernel = int(values.get("vcm-kernel", o.kernel))

    def flag(key, bit):
        nonlocal o
        current = bool(o.options & bit)
        if key == "vcm-merge_vertices":  # merge_vertices() = enable_merging() && MergeVertices (vcm_shared.hxx:58-60)
            current = bool(o.options & api.VCM_ENABLE_MERGING) and current
        enabled = bool(values.get(key, current))
        o.options = (o.options | bit) if enabled else (o.options & ~bit)

    flag("vcm-direct_hit", api.VCM_DIRECT_HIT)
    flag("vcm-connect_to_light", api.VCM_CONNECT_TO_LIGHT)
    flag("vcm-connect_to_camera", api.VCM_CONNECT_TO_CAMERA)
    flag("vcm-connect_vertices", api.VCM_CONNECT_VERTICES)
    flag("vcm-merge_vertices", api.VCM_MERGE_VERTICES)
    flag("vcm-mis", api.VCM_ENABLE_MIS)
    flag("vcm-merging", api.VCM_ENABLE_MERGING)
    # the backend's own key (integration/etx_hip_integrators.hxx): camera path i keeps the sampler state of light path i (vcm_shared.hxx:312,357)
    o.reference_seeding = 1 if values.get("hip-reference_seeding", False) else 0
    return o


def pt_options_from_dict(values):
    """CPUPathTracingImpl::start (sources/etx/rt/integrators/path_tracing.cxx:36-40): keys direct / nee / mis / bn."""
    o = api.PTOptions.default_values()
    o.direct = 1 if values.get("direct", bool(o.direct)) else 0
    o.nee = 1 if values.get("nee", bool(o.nee)) else 0
    o.mis = 1 if values.get("mis", bool(o.mis)) else 0
    o.blue_noise = 1 if values.get("bn", bool(o.blue_noise)) else 0
    return o


def bdpt_options_from_dict(values):
    """CPUBidirectionalImpl::start (sources/etx/rt/integrators/bidirectional.cxx:1469-1478): same keys, same defaults."""
    o = api.BDPTOptions.default_values()
    o.mode = int(values.get("bdpt-mode", o.mode))
    for key, field in (("bdpt-conn_direct_hit", "direct_hit"), ("bdpt-conn_connect_to_camera", "connect_to_camera"), ("bdpt-conn_connect_to_light", "connect_to_light"),
                       ("bdpt-conn_connect_vertices", "connect_vertices"), ("bdpt-conn_mis", "mis"), ("bdpt-blue_noise", "blue_noise")):
        setattr(o, field, 1 if values.get(key, bool(getattr(o, field))) else 0)
    o.reference_seeding = 1 if values.get("hip-reference_seeding", False) else 0  # bidirectional.cxx:377-378
    return o


class HIPIntegrator(Integrator):
    """run / update / stop protocol shared by the device integrators (one iteration per update())."""

    def __init__(self, snapshot, device=0, first_iteration=0, iteration_stride=1, pixel_first=0, pixel_stride=1):
        super().__init__()
        self.snapshot = snapshot
        self.context = api.Context(device)
        self.first_iteration = first_iteration
        self.iteration_stride = iteration_stride
        # pixel-interleaved sharding (etx_hip_begin_ex; path tracer and bidirectional integrator): this rank's film is a zero-padded part of the job's
        self.pixel_first = pixel_first
        self.pixel_stride = pixel_stride
        self._uploaded_version = None  # snapshot.version at the last etx_hip_upload_scene
        self._pending_changes = None   # scene_edited(): what the host changed since (None: unknown = upload everything)
        self.bvh_builder = api.BVH_HOST_SAH  # api.BVH_DEVICE_LBVH: the tree of the next upload is built on the device (etx_hip_set_bvh_builder)
        self._rendered = 0
        self._have_camera_image = False
        self._have_light_image = False
        # what the C++ host tabulates from its BNSampler (include/etx_hip.h): {set_index: uint8 [128,128,256,8]}
        self.bluenoise_tables = {}
        self.cie_table = None  # spectral scenes: (float32 [count, 3] spectrum::spectral_xyz, first wavelength) of the host
        self.rgb_response_table = None  # spectral scenes with RGB images: (float32 [count, 3] rgb_response rows, first wavelength) of the host
        self._bluenoise_uploaded = set()

    def _begin(self):
        raise NotImplementedError

    def _iterations_to_render(self):
        # this rank renders first, first + stride, ... < scene.samples (vcm_cpu.cxx:234: stop at iteration + 1 >= samples)
        total = self.snapshot.samples
        if self.first_iteration >= total:
            return 0
        return (total - self.first_iteration + self.iteration_stride - 1) // self.iteration_stride

    def scene_edited(self, changed):
        """The host says what it edited in the snapshot since the last run (api.CHANGED_*): the next run() updates the device scene in
        place (etx_hip_update_scene: no BVH rebuild, no image upload) instead of uploading it again. Without this call an
        edited snapshot (snapshot.version) is uploaded as a whole - what the reference does on every change (app.cxx:364-403)."""
        self._pending_changes = (self._pending_changes or 0) | int(changed)

    def run(self):
        self.stop(Stop.Immediate)
        self.context.set_bvh_builder(self.bvh_builder)
        if (self._uploaded_version is not None) and (self._pending_changes is not None):
            self.context.update_scene(self.snapshot, self._pending_changes)
            self._uploaded_version = self.snapshot.version
        elif self._uploaded_version != self.snapshot.version:  # first run, or the host edited the scene since (app.cxx:364-403 restarts)
            self.context.upload_scene(self.snapshot)
            self._uploaded_version = self.snapshot.version
        self._pending_changes = None
        if self.cie_table is not None:
            self.context.upload_cie_table(*self.cie_table)
        if self.rgb_response_table is not None:
            self.context.upload_rgb_response(*self.rgb_response_table)
        for set_index, table in self.bluenoise_tables.items():
            if set_index not in self._bluenoise_uploaded:
                self.context.upload_bluenoise(set_index, table)
                self._bluenoise_uploaded.add(set_index)
        self._begin()
        self._rendered = 0
        self.current_state = State.Running if self._iterations_to_render() > 0 else State.Stopped

    def update(self):
        if self.current_state == State.Stopped:
            return
        # asynchronous and non-blocking (vcm_cpu.cxx:264-268: update() returns at once while work is in flight): the iteration
        # goes to a free device lane, or nowhere when every lane is busy - the next update() tries again
        if self.context.try_render_iteration() == 0:
            return
        self._rendered += 1
        self._have_camera_image = True
        self._have_light_image = True
        if (self.current_state == State.WaitingForCompletion) or (self._rendered >= self._iterations_to_render()):
            self.context.sync()
            self.current_state = State.Stopped

    def stop(self, how=Stop.Immediate):
        if self.current_state == State.Stopped:
            return
        if how == Stop.Immediate:
            self.context.sync()
            self.current_state = State.Stopped
        else:
            self.current_state = State.WaitingForCompletion

    def have_updated_camera_image(self):
        r, self._have_camera_image = self._have_camera_image, False
        return r

    def have_updated_light_image(self):
        r, self._have_light_image = self._have_light_image, False
        return r

    def status(self):
        return self.context.stats()

    def render(self):
        """run() + update() until Stopped - what the headless driver does (oracle/driver/etx_oracle.cxx main loop)."""
        self.run()
        return self.finish()

    def finish(self):
        """update() until Stopped (after run() or resume())."""
        while self.state() != State.Stopped:
            before = self._rendered
            self.update()
            if (self._rendered == before) and (self.state() != State.Stopped):
                time.sleep(1.0e-4)  # every lane busy: the GUI would come back at its next frame (driver: sleep_for(200us))
        return self

    def film(self, layer=api.LAYER_RESULT):
        return self.context.read_film(layer)

    def save_checkpoint(self, path=None):
        """The film state of the render in progress (waits for the iterations in flight; rendering may go on afterwards).
        The reference has no checkpoint: this is what a headless driver stores next to its output image."""
        blob = self.context.checkpoint_save()
        if path is not None:
            with open(path, "wb") as f:
                f.write(blob)
        return blob

    def resume(self, checkpoint):
        """run() continued from a checkpoint (bytes or a file name) of the same scene, integrator, options and sharding: the
        iterations the saved render had completed are not rendered again, the remaining ones carry their own indices."""
        if isinstance(checkpoint, str):
            with open(checkpoint, "rb") as f:
                checkpoint = f.read()
        self.run()
        self.context.checkpoint_load(checkpoint)
        self._rendered = int(self.context.stats().completed_iterations)
        self.current_state = State.Running if self._rendered < self._iterations_to_render() else State.Stopped
        return self


class HIPVCM(HIPIntegrator):
    """CPUVCM (sources/etx/rt/integrators/vcm_cpu.cxx:243-310) on the device."""

    def name(self):
        return "VCM (HIP gfx950)"

    def _begin(self):
        if self.pixel_stride != 1:  # refused by the library with its reason (the photon map needs every pixel's light path)
            self.context.begin_ex(api.INTEGRATOR_VCM, vcm_options_from_dict(self.integrator_options), self.first_iteration, self.iteration_stride, self.pixel_first, self.pixel_stride)
            return
        self.context.begin_vcm(vcm_options_from_dict(self.integrator_options), self.first_iteration, self.iteration_stride)


class HIPPathTracing(HIPIntegrator):
    """CPUPathTracing (sources/etx/rt/integrators/path_tracing.cxx:112-172) on the device. film(LAYER_NORMAL / LAYER_ALBEDO)
    are the AOVs Film::accumulate_camera_image receives."""

    def name(self):
        return "Path Tracing (HIP gfx950)"

    def _begin(self):
        self.context.begin_pt(pt_options_from_dict(self.integrator_options), self.first_iteration, self.iteration_stride, self.pixel_first, self.pixel_stride)

    def update(self):
        super().update()
        # CPUPathTracingImpl::update (path_tracing.cxx:91-93): an iteration that sampled no pixel - every pixel has converged
        # (Film::estimate_noise_levels) - ends the render. Iterations already in flight on the device lanes finish as no-ops.
        if (self.current_state == State.Running) and (self.snapshot.noise_threshold > 0.0) and (self._rendered > 33):
            stats = self.context.stats()
            if (stats.completed_iterations > 33) and (stats.last_active_pixels == 0):
                self.stop(Stop.WaitForCompletion)
                self.context.sync()
                self.current_state = State.Stopped


class HIPBidirectional(HIPIntegrator):
    """CPUBidirectional (sources/etx/rt/integrators/bidirectional.cxx:1490-1560) on the device, all four bdpt-mode values;
    see include/etx_hip.h for what etx_hip_begin rejects (random-walk subsurface scenes)."""

    def name(self):
        return "Bidirectional (HIP gfx950)"

    def _begin(self):
        self.context.begin_bdpt(bdpt_options_from_dict(self.integrator_options), self.first_iteration, self.iteration_stride, self.pixel_first, self.pixel_stride)
