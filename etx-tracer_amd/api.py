"""ctypes binding of libetx_hip.so (include/etx_hip.h). One Python method per C entry point, same names."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

LAYER_CAMERA, LAYER_LIGHT, LAYER_RESULT, LAYER_NORMAL, LAYER_ALBEDO = 0, 1, 2, 3, 4
INTEGRATOR_PT, INTEGRATOR_VCM, INTEGRATOR_BDPT = 0, 1, 2
BDPT_MODE_PATH_TRACING, BDPT_MODE_LIGHT_TRACING, BDPT_MODE_FAST, BDPT_MODE_FULL = 0, 1, 2, 3

# etx::VCMOptions bits (sources/etx/rt/shared/vcm_shared.hxx:24-37)
VCM_CONNECT_TO_CAMERA = 1 << 0
VCM_DIRECT_HIT = 1 << 1
VCM_CONNECT_TO_LIGHT = 1 << 2
VCM_CONNECT_VERTICES = 1 << 3
VCM_MERGE_VERTICES = 1 << 4
VCM_ENABLE_MIS = 1 << 5
VCM_ENABLE_MERGING = 1 << 6
VCM_FULL_OPTIONS = 0x7F
CHANGED_CAMERA, CHANGED_MATERIALS, CHANGED_POSITIONS, REBUILD_BVH = 1, 2, 4, 8  # etx_hip_update_scene
BVH_HOST_SAH, BVH_DEVICE_LBVH = 0, 1  # etx_hip_set_bvh_builder
REDUCE_SUM, REDUCE_MAX, REDUCE_MIN = 0, 1, 2  # etx_hip_comm_all_reduce_f64

EXPORTED_SYMBOLS = (
    "etx_hip_abi_version", "etx_hip_create", "etx_hip_destroy", "etx_hip_last_error", "etx_hip_upload_scene", "etx_hip_update_scene",
    "etx_hip_upload_bluenoise", "etx_hip_upload_cie_table", "etx_hip_upload_rgb_response", "etx_hip_begin", "etx_hip_begin_ex", "etx_hip_render_iteration", "etx_hip_try_render_iteration", "etx_hip_poll", "etx_hip_sync",
    "etx_hip_read_film", "etx_hip_read_film_begin", "etx_hip_read_film_end", "etx_hip_checkpoint_bytes", "etx_hip_checkpoint_save", "etx_hip_checkpoint_load", "etx_hip_stats", "etx_hip_set_timers", "etx_hip_set_debug_flags", "etx_hip_set_pool_policy", "etx_hip_lanes", "etx_hip_device_bytes", "etx_hip_comm_unique_id", "etx_hip_comm_init", "etx_hip_reduce_film", "etx_hip_reduce_film_begin", "etx_hip_reduce_film_end", "etx_hip_reduce_info",
    "etx_hip_trace_rays", "etx_hip_trace_rays_device", "etx_hip_kat", "etx_hip_host_check_bvh", "etx_hip_host_bvh_stats",
    "etx_hip_set_bvh_builder", "etx_hip_bvh_info", "etx_hip_selftest_stack", "etx_hip_host_check_bvh_builder", "etx_hip_host_bvh_stats_builder",
    "etx_hip_runtime_info", "etx_hip_comm_all_reduce_f64", "etx_hip_comm_barrier", "etx_hip_trace_rays_timed",
)


class EtxHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("etx_hip error %d: %s" % (code, message))
        self.code = code


class VCMOptions(ctypes.Structure):
    """etx_abi_vcm_options == etx::VCMOptions (vcm_shared.hxx:12-72), 32 bytes, 16-byte aligned."""
    _fields_ = [
        ("options", ctypes.c_uint32),
        ("radius_decay", ctypes.c_uint32),
        ("kernel", ctypes.c_uint32),
        ("initial_radius", ctypes.c_float),
        ("blue_noise", ctypes.c_uint8),
        ("reference_seeding", ctypes.c_uint8),  # in VCMOptions' tail padding: option key "hip-reference_seeding" (etx_scene_abi.h)
        ("_pad", ctypes.c_uint8 * 14),
    ]

    @staticmethod
    def default_values():
        """VCMOptions::default_values (sources/etx/rt/integrators/vcm_shared.cxx:6-13) + blue_noise default true."""
        o = VCMOptions()
        o.options = VCM_FULL_OPTIONS
        o.radius_decay = 256
        o.kernel = 1  # Epanechnikov
        o.initial_radius = 0.0
        o.blue_noise = 1
        return o


class PTOptions(ctypes.Structure):
    """etx_abi_pt_options == etx::PTOptions (path_tracing_shared.hxx:8-14), 16 bytes."""
    _fields_ = [
        ("path_per_iteration", ctypes.c_uint32),
        ("nee", ctypes.c_uint8),
        ("direct", ctypes.c_uint8),
        ("mis", ctypes.c_uint8),
        ("blue_noise", ctypes.c_uint8),
        ("_pad", ctypes.c_uint8 * 8),
    ]

    @staticmethod
    def default_values():
        o = PTOptions()
        o.path_per_iteration = 1
        o.nee = o.direct = o.mis = o.blue_noise = 1
        return o


class BDPTOptions(ctypes.Structure):
    """etx_abi_bdpt_options: CPUBidirectionalImpl's option members (bidirectional.cxx:323-340, 1443-1466), 16 bytes."""
    _fields_ = [
        ("mode", ctypes.c_uint32),
        ("direct_hit", ctypes.c_uint8),
        ("connect_to_camera", ctypes.c_uint8),
        ("connect_to_light", ctypes.c_uint8),
        ("connect_vertices", ctypes.c_uint8),
        ("mis", ctypes.c_uint8),
        ("blue_noise", ctypes.c_uint8),
        ("reference_seeding", ctypes.c_uint8),
        ("_pad", ctypes.c_uint8 * 5),
    ]

    @staticmethod
    def default_values():
        o = BDPTOptions()
        o.mode = BDPT_MODE_FAST  # CPUBidirectionalImpl::mode, bidirectional.cxx:332
        o.direct_hit = o.connect_to_camera = o.connect_to_light = o.connect_vertices = o.mis = o.blue_noise = 1
        return o


class ReduceInfo(ctypes.Structure):
    """etx_hip_reduce_info_t"""
    _fields_ = [
        ("reduces", ctypes.c_uint64),
        ("payload_bytes", ctypes.c_uint64),
        ("global_iterations", ctypes.c_uint64),
        ("last_device_ms", ctypes.c_double),
        ("total_device_ms", ctypes.c_double),
        ("pending", ctypes.c_uint32),
        ("layer_mask", ctypes.c_uint32),
    ]


class Stats(ctypes.Structure):
    """etx_hip_stats_t"""
    _fields_ = [
        ("last_iteration_time", ctypes.c_double),
        ("total_time", ctypes.c_double),
        ("completed_iterations", ctypes.c_uint32),
        ("current_iteration", ctypes.c_uint32),
        ("rays_extension", ctypes.c_uint64),
        ("rays_shadow", ctypes.c_uint64),
        ("light_vertices", ctypes.c_uint64),
        ("camera_vertices", ctypes.c_uint64),
        ("photons_examined", ctypes.c_uint64),
        ("photons_merged", ctypes.c_uint64),
        ("splats", ctypes.c_uint64),
        ("wavefront_bounces", ctypes.c_uint64),
        ("overflow_flags", ctypes.c_uint32),
        ("nonfinite_dropped", ctypes.c_uint32),
        ("ms_trace_closest", ctypes.c_double),
        ("ms_trace_shadow", ctypes.c_double),
        ("ms_shade_light", ctypes.c_double),
        ("ms_shade_camera", ctypes.c_double),
        ("ms_connect", ctypes.c_double),
        ("ms_merge", ctypes.c_double),
        ("ms_grid_build", ctypes.c_double),
        ("ms_generate", ctypes.c_double),
        ("launches_trace_closest", ctypes.c_uint64),
        ("launches_trace_shadow", ctypes.c_uint64),
        ("rays_light", ctypes.c_uint64),
        ("rays_camera", ctypes.c_uint64),
        ("pairs", ctypes.c_uint64),
        ("endpoints", ctypes.c_uint64),
        ("active_pixels", ctypes.c_uint64),
        ("last_active_pixels", ctypes.c_uint64),
        ("boundary_crossings", ctypes.c_uint64),
        ("pool_grows", ctypes.c_uint32),
        ("reserved_0", ctypes.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "pad"}


def library_path():
    return os.environ.get("ETX_HIP_LIBRARY", os.path.join(_HERE, "libetx_hip.so"))


class Library:
    """Loads libetx_hip.so. Raises if it is missing: the product path never falls back to a CPU implementation."""

    _instance = None

    def __init__(self, path=None):
        path = path or library_path()
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(etx-tracer_amd has no CPU fallback)" % path)
        self.path = path
        # RTLD_NOW: the library and the ROCm runtime it pulls in (/opt/rocm: libamdhip64, libhsa-runtime64, librccl - none of them linked -z now) bind
        # every symbol HERE. With lazy binding a package imported later that maps a second ROCm stack into the global namespace (the torch wheel
        # bundles 7.0.2 under torch/lib) would capture whichever calls had not happened yet - seen in round 6: hipRuntimeGetVersion answered by the
        # wheel's runtime in a process whose libetx_hip.so had been loaded against /opt/rocm first.
        # (ETX_HIP_DLOPEN_LAZY=1: the default lazy binding again - a diagnostic of round 6, tools/gpu_calls/gpu_r6g.sh)
        self.lib = ctypes.CDLL(path) if os.environ.get("ETX_HIP_DLOPEN_LAZY") else ctypes.CDLL(path, mode=os.RTLD_NOW | os.RTLD_LOCAL)
        L = self.lib
        vp, u32, u64, i32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_size_t
        L.etx_hip_abi_version.restype = i32
        L.etx_hip_create.argtypes = [i32, ctypes.POINTER(vp)]
        L.etx_hip_destroy.argtypes = [vp]
        L.etx_hip_destroy.restype = None
        L.etx_hip_last_error.argtypes = [vp]
        L.etx_hip_last_error.restype = ctypes.c_char_p
        L.etx_hip_upload_scene.argtypes = [vp, vp, vp]
        L.etx_hip_update_scene.argtypes = [vp, vp, vp, u32]
        L.etx_hip_upload_bluenoise.argtypes = [vp, u32, vp, sz]
        L.etx_hip_upload_cie_table.argtypes = [vp, vp, u32, ctypes.c_float]
        L.etx_hip_upload_rgb_response.argtypes = [vp, vp, u32, ctypes.c_float]
        L.etx_hip_begin.argtypes = [vp, i32, vp, sz, u32, u32]
        L.etx_hip_begin_ex.argtypes = [vp, i32, vp, sz, u32, u32, u32, u32]
        L.etx_hip_render_iteration.argtypes = [vp]
        L.etx_hip_try_render_iteration.argtypes = [vp]
        L.etx_hip_poll.argtypes = [vp]
        L.etx_hip_sync.argtypes = [vp]
        L.etx_hip_read_film.argtypes = [vp, i32, vp, sz]
        L.etx_hip_read_film_begin.argtypes = [vp, i32]
        L.etx_hip_read_film_end.argtypes = [vp, vp, sz, i32]
        L.etx_hip_checkpoint_bytes.argtypes = [vp]
        L.etx_hip_checkpoint_bytes.restype = sz
        L.etx_hip_checkpoint_save.argtypes = [vp, vp, sz]
        L.etx_hip_checkpoint_load.argtypes = [vp, vp, sz]
        L.etx_hip_stats.argtypes = [vp, ctypes.POINTER(Stats), sz]
        L.etx_hip_set_timers.argtypes = [vp, u32]
        L.etx_hip_set_debug_flags.argtypes = [vp, u32]
        L.etx_hip_set_pool_policy.argtypes = [vp, u32, sz]
        L.etx_hip_lanes.argtypes = [vp, ctypes.c_int]
        L.etx_hip_lanes.restype = u32
        L.etx_hip_device_bytes.argtypes = [vp]
        L.etx_hip_device_bytes.restype = ctypes.c_size_t
        L.etx_hip_comm_unique_id.argtypes = [vp]
        L.etx_hip_comm_init.argtypes = [vp, i32, i32, vp]
        L.etx_hip_reduce_film.argtypes = [vp]
        L.etx_hip_reduce_film_begin.argtypes = [vp]
        L.etx_hip_reduce_film_end.argtypes = [vp, i32]
        L.etx_hip_reduce_info.argtypes = [vp, vp, ctypes.c_size_t]
        L.etx_hip_trace_rays.argtypes = [vp, vp, u64, vp]
        L.etx_hip_trace_rays_device.argtypes = [vp, vp, vp, u64, vp, u32, ctypes.POINTER(ctypes.c_double)]
        if hasattr(L, "etx_hip_runtime_info"):  # ABI 4. (A library of ABI 3 - tools/gpu_calls A/B runs against etx-tracer_amd/variants/libetx_hip_r5.so - still loads.)
            L.etx_hip_trace_rays_timed.argtypes = [vp, vp, u64, u32, ctypes.POINTER(ctypes.c_double), vp]
            L.etx_hip_runtime_info.argtypes = [ctypes.POINTER(i32 * 4)]
            L.etx_hip_comm_all_reduce_f64.argtypes = [vp, ctypes.POINTER(ctypes.c_double), u32, i32]
            L.etx_hip_comm_barrier.argtypes = [vp]
        L.etx_hip_kat.argtypes = [vp, i32, vp, u64, vp]
        L.etx_hip_host_check_bvh.argtypes = [vp, ctypes.POINTER(u32 * 4)]
        L.etx_hip_host_bvh_stats.argtypes = [vp, vp, u64, ctypes.POINTER(u64 * 4)]
        L.etx_hip_set_bvh_builder.argtypes = [vp, i32]
        L.etx_hip_selftest_stack.argtypes = [vp, u32, ctypes.POINTER(u32)]
        L.etx_hip_bvh_info.argtypes = [vp, ctypes.POINTER(u32 * 4), ctypes.POINTER(ctypes.c_double)]
        L.etx_hip_host_check_bvh_builder.argtypes = [vp, i32, ctypes.POINTER(u32 * 4)]
        L.etx_hip_host_bvh_stats_builder.argtypes = [vp, i32, vp, u64, ctypes.POINTER(u64 * 4), vp]

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = Library()
        return cls._instance

    def has_symbol(self, name):
        return hasattr(self.lib, name)


class Context:
    """RAII wrapper of etx_hip_context."""

    def __init__(self, device=0, library=None):
        self.library = library or Library.get()
        self.handle = ctypes.c_void_p()
        rc = self.library.lib.etx_hip_create(int(device), ctypes.byref(self.handle))
        if rc != 0:
            msg = self.library.lib.etx_hip_last_error(None)
            self.handle = None
            raise EtxHipError(rc, msg.decode() if msg else "")
        self.film_size = None
        self._scene_keepalive = None

    def close(self):
        if self.handle:
            self.library.lib.etx_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            msg = self.library.lib.etx_hip_last_error(self.handle)
            raise EtxHipError(rc, msg.decode() if msg else "")
        return rc

    def upload_scene(self, snapshot):
        self._check(self.library.lib.etx_hip_upload_scene(self.handle, snapshot.scene_address, snapshot.camera_address))
        self.film_size = snapshot.film_size
        self._scene_keepalive = snapshot

    def selftest_stack(self, depth):
        """-> mismatches of the traversal stack round trip at `depth` entries per lane (etx_hip_selftest_stack); 0 expected"""
        errors = ctypes.c_uint32(0xFFFFFFFF)
        self._check(self.library.lib.etx_hip_selftest_stack(self.handle, int(depth), ctypes.byref(errors)))
        return errors.value

    def set_bvh_builder(self, builder):
        """BVH_HOST_SAH (default) or BVH_DEVICE_LBVH: who builds the tree of the next upload_scene (etx_hip_set_bvh_builder)."""
        self._check(self.library.lib.etx_hip_set_bvh_builder(self.handle, int(builder)))

    def bvh_info(self):
        """{nodes, triangles, depth, stack_need, bytes, build_ms} of the uploaded scene's traversal tree."""
        info = (ctypes.c_uint32 * 4)()
        build_ms = ctypes.c_double(0.0)
        self._check(self.library.lib.etx_hip_bvh_info(self.handle, ctypes.byref(info), ctypes.byref(build_ms)))
        return {"nodes": info[0], "triangles": info[1], "depth": info[2] & 0xffff, "stack_need": info[2] >> 16, "bytes": info[3], "build_ms": build_ms.value}

    def update_scene(self, snapshot, changed):
        """The host edited the uploaded scene in place: CHANGED_CAMERA | CHANGED_MATERIALS | CHANGED_POSITIONS (etx_hip_update_scene:
        tables rebuilt, geometry / BVH / images stay on the device, moved vertices refit the BVH there)."""
        self._check(self.library.lib.etx_hip_update_scene(self.handle, snapshot.scene_address, snapshot.camera_address, int(changed)))
        self._scene_keepalive = snapshot

    def upload_bluenoise(self, set_index, values):
        """values: uint8 array [128,128,256,8] - the host's sample_blue_noise outputs for one sample-count class."""
        import numpy as np
        table = np.ascontiguousarray(values, dtype=np.uint8)
        self._check(self.library.lib.etx_hip_upload_bluenoise(self.handle, int(set_index), table.ctypes.data, table.nbytes))

    def upload_cie_table(self, xyz, first_wavelength):
        """xyz: float32 [count, 3] = spectrum::spectral_xyz(i) of the host, 1 nm apart from first_wavelength (spectral scenes)."""
        table = np.ascontiguousarray(xyz, dtype=np.float32)
        self._check(self.library.lib.etx_hip_upload_cie_table(self.handle, table.ctypes.data, table.shape[0], float(first_wavelength)))

    def upload_rgb_response(self, rgb, first_wavelength):
        """rgb_response of the unit colours per integer wavelength (float32 [count, 3]): spectral scenes with RGB images."""
        table = np.ascontiguousarray(rgb, dtype=np.float32)
        self._check(self.library.lib.etx_hip_upload_rgb_response(self.handle, table.ctypes.data, table.shape[0], float(first_wavelength)))

    def begin_vcm(self, options, first_iteration=0, iteration_stride=1):
        self._check(self.library.lib.etx_hip_begin(self.handle, INTEGRATOR_VCM, ctypes.byref(options), ctypes.sizeof(options), first_iteration, iteration_stride))

    def begin_pt(self, options, first_iteration=0, iteration_stride=1, pixel_first=0, pixel_stride=1):
        self.begin_ex(INTEGRATOR_PT, options, first_iteration, iteration_stride, pixel_first, pixel_stride)

    def begin_bdpt(self, options, first_iteration=0, iteration_stride=1, pixel_first=0, pixel_stride=1):
        self.begin_ex(INTEGRATOR_BDPT, options, first_iteration, iteration_stride, pixel_first, pixel_stride)

    def begin_ex(self, integrator, options, first_iteration=0, iteration_stride=1, pixel_first=0, pixel_stride=1):
        """etx_hip_begin_ex: iteration sharding and, for the path tracer / bidirectional integrator, pixel-interleaved sharding."""
        self._check(self.library.lib.etx_hip_begin_ex(self.handle, int(integrator), ctypes.byref(options), ctypes.sizeof(options), first_iteration, iteration_stride,
                                                     pixel_first, pixel_stride))

    def render_iteration(self):
        self._check(self.library.lib.etx_hip_render_iteration(self.handle))

    def lanes(self, integrator=INTEGRATOR_VCM):
        """Iterations in flight under that integrator (etx_hip_lanes)."""
        return int(self.library.lib.etx_hip_lanes(self.handle, int(integrator)))

    def device_bytes(self):
        """Device memory of the lanes' working sets in bytes (etx_hip_device_bytes)."""
        return int(self.library.lib.etx_hip_device_bytes(self.handle))

    def set_pool_policy(self, initial_light_vertices_per_path=0, max_pool_bytes_per_lane=0):
        """Start size (light vertices per path, 0 = default) and byte limit (0 = none) of a lane's growable pools, for the next upload_scene."""
        self._check(self.library.lib.etx_hip_set_pool_policy(self.handle, int(initial_light_vertices_per_path), int(max_pool_bytes_per_lane)))

    def set_debug_flags(self, flags):
        """Ablation switches of the kernels (etx_hip_set_debug_flags; 0 = production): timing experiments and kernel-level tests."""
        self._check(self.library.lib.etx_hip_set_debug_flags(self.handle, int(flags)))

    def set_timers(self, mask):
        """Kernel groups timed with HIP events (bit i = i-th ms_* field of the stats, 0xff = all)."""
        self._check(self.library.lib.etx_hip_set_timers(self.handle, int(mask)))

    def try_render_iteration(self):
        """1 = handed to a free lane, 0 = every lane busy; never blocks (Integrator::update must not block)."""
        return self._check(self.library.lib.etx_hip_try_render_iteration(self.handle))

    def poll(self):
        return self._check(self.library.lib.etx_hip_poll(self.handle))

    def sync(self):
        self._check(self.library.lib.etx_hip_sync(self.handle))

    def read_film(self, layer):
        w, h = self.film_size
        out = np.empty((h, w, 4), dtype=np.float32)
        self._check(self.library.lib.etx_hip_read_film(self.handle, layer, out.ctypes.data, out.nbytes))
        return out

    def read_film_begin(self, layer):
        """Asynchronous read-back: never waits for iterations in flight (etx_hip_read_film_begin)."""
        self._check(self.library.lib.etx_hip_read_film_begin(self.handle, layer))

    def read_film_end(self, wait=False):
        """-> the image, or None while the copy is still running (wait=False never blocks)."""
        w, h = self.film_size
        out = np.empty((h, w, 4), dtype=np.float32)
        rc = self._check(self.library.lib.etx_hip_read_film_end(self.handle, out.ctypes.data, out.nbytes, 1 if wait else 0))
        return out if rc == 1 else None

    def checkpoint_save(self):
        """-> bytes: the film state of the render in progress (etx_hip_checkpoint_save; waits for the iterations in flight)."""
        size = self.library.lib.etx_hip_checkpoint_bytes(self.handle)
        if size == 0:
            raise EtxHipError(-1, "checkpoint_save: call begin() first")
        buffer = np.empty(size, dtype=np.uint8)
        self._check(self.library.lib.etx_hip_checkpoint_save(self.handle, buffer.ctypes.data, buffer.nbytes))
        return buffer.tobytes()

    def checkpoint_load(self, blob):
        """After begin() with the same integrator, options and sharding: continue the saved render (etx_hip_checkpoint_load)."""
        buffer = np.frombuffer(blob, dtype=np.uint8)
        self._check(self.library.lib.etx_hip_checkpoint_load(self.handle, buffer.ctypes.data, buffer.nbytes))

    def stats(self):
        s = Stats()
        self._check(self.library.lib.etx_hip_stats(self.handle, ctypes.byref(s), ctypes.sizeof(s)))
        return s

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.empty((rays.shape[0], 4), dtype=np.float32)
        self._check(self.library.lib.etx_hip_trace_rays(self.handle, rays.ctypes.data, rays.shape[0], hits.ctypes.data))
        return hits

    def trace_rays_device(self, d_o_tmin, d_d_tmax, count, d_hits, repeat):
        ms = ctypes.c_double()
        self._check(self.library.lib.etx_hip_trace_rays_device(self.handle, d_o_tmin, d_d_tmax, count, d_hits, repeat, ctypes.byref(ms)))
        return ms.value

    def trace_rays_timed(self, rays, repeat, want_hits=False):
        """etx_hip_trace_rays_timed: host rays [n, 8] uploaded once, one untimed + `repeat` timed launches of the production traversal kernel over the
        device-resident queue -> average kernel time in ms (HIP events on the launch stream) [, hits of the last launch]."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.empty((rays.shape[0], 4), dtype=np.float32) if want_hits else None
        ms = ctypes.c_double()
        self._check(self.library.lib.etx_hip_trace_rays_timed(self.handle, rays.ctypes.data, rays.shape[0], int(repeat), ctypes.byref(ms), hits.ctypes.data if want_hits else None))
        return (ms.value, hits) if want_hits else ms.value

    def kat(self, which, values, out_width):
        values = np.ascontiguousarray(values, dtype=np.float32)
        count = values.shape[0]
        out = np.empty((count, out_width), dtype=np.float32)
        self._check(self.library.lib.etx_hip_kat(self.handle, which, values.ctypes.data, count, out.ctypes.data))
        return out

    def comm_init(self, rank, world, unique_id_bytes):
        buf = ctypes.create_string_buffer(bytes(unique_id_bytes), 128)
        self._check(self.library.lib.etx_hip_comm_init(self.handle, rank, world, buf))

    def comm_all_reduce(self, values, op=REDUCE_SUM):
        """etx_hip_comm_all_reduce_f64: up to 16 doubles reduced over the context's communicator (identity without one). -> list of floats"""
        values = [float(v) for v in values]
        words = (ctypes.c_double * len(values))(*values)
        self._check(self.library.lib.etx_hip_comm_all_reduce_f64(self.handle, words, len(values), int(op)))
        return [float(w) for w in words]

    def comm_barrier(self):
        self._check(self.library.lib.etx_hip_comm_barrier(self.handle))

    def reduce_film(self):
        """etx_hip_sync + one film reduce, blocking; rendering may continue afterwards."""
        self._check(self.library.lib.etx_hip_reduce_film(self.handle))

    def reduce_film_begin(self):
        """Enqueues a film reduce (snapshot + out-of-place all-reduce) on the communication stream and returns; the lanes keep rendering."""
        self._check(self.library.lib.etx_hip_reduce_film_begin(self.handle))

    def reduce_film_end(self, wait=True):
        """True once the reduced copy is complete (False while it is still running, wait=False only)."""
        rc = self.library.lib.etx_hip_reduce_film_end(self.handle, 1 if wait else 0)
        if rc < 0:
            self._check(rc)
        return rc == 1

    def reduce_info(self):
        info = ReduceInfo()
        self._check(self.library.lib.etx_hip_reduce_info(self.handle, ctypes.byref(info), ctypes.sizeof(info)))
        return info


def host_check_bvh(snapshot, library=None, builder=BVH_HOST_SAH):
    """(rc, {nodes, triangles, depth, bytes}) - host-only BVH build + invariant check, no GPU needed. builder=BVH_DEVICE_LBVH: the
    device build, emulated on the host through the functions its kernels call."""
    library = library or Library.get()
    info = (ctypes.c_uint32 * 4)()
    rc = library.lib.etx_hip_host_check_bvh_builder(snapshot.scene_address, int(builder), ctypes.byref(info))
    return rc, {"nodes": info[0], "triangles": info[1], "depth": info[2] & 0xffff, "stack_need": info[2] >> 16, "bytes": info[3]}


def host_bvh_stats(snapshot, rays, library=None, builder=BVH_HOST_SAH, with_hits=False):
    """Host-only: the work the traversal does for `rays` (n x 8 float32): node visits, triangle tests, hits, deepest stack use;
    with_hits: + "t" (float32 n) and "triangle" (int64 n, -1 = miss) of every ray's closest hit."""
    library = library or Library.get()
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = (ctypes.c_uint64 * 4)()
    hits = np.zeros((rays.shape[0], 2), dtype=np.float32) if with_hits else None
    rc = library.lib.etx_hip_host_bvh_stats_builder(snapshot.scene_address, int(builder), rays.ctypes.data, rays.shape[0], ctypes.byref(out), hits.ctypes.data if with_hits else None)
    result = {"node_visits": out[0], "triangle_tests": out[1], "hits": out[2], "max_stack": out[3]}
    if with_hits:
        triangle = hits[:, 1].view(np.uint32).astype(np.int64)
        triangle[triangle == 0xFFFFFFFF] = -1
        result["t"], result["triangle"] = hits[:, 0].copy(), triangle
    return rc, result


def runtime_info(library=None):
    """{hip_runtime, hip_built_against, rccl_runtime, rccl_built_against, mapped}: the ROCm runtime the loader bound libetx_hip.so to in THIS process
    (etx_hip_runtime_info) and the files it is mapped from. One libamdhip64 / libhsa-runtime64 / librccl each is what a healthy process shows."""
    library = library or Library.get()
    versions = (ctypes.c_int * 4)()
    library.lib.etx_hip_runtime_info(ctypes.byref(versions))
    mapped = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1] if "/" in line else ""
                if any(k in path for k in ("libamdhip64", "libhsa-runtime64", "librccl")) and path not in mapped:
                    mapped.append(path)
    except OSError:
        pass
    return {"hip_runtime": versions[0], "hip_built_against": versions[1], "rccl_runtime": versions[2], "rccl_built_against": versions[3], "mapped": mapped}


def comm_unique_id(library=None):
    library = library or Library.get()
    buf = ctypes.create_string_buffer(128)
    rc = library.lib.etx_hip_comm_unique_id(buf)
    if rc != 0:
        raise EtxHipError(rc, "etx_hip_comm_unique_id failed")
    return bytes(buf.raw)
