"""TEST DOUBLE (not product code): stands in for etx_tracer_amd.api.Context where no GPU exists, so that bench.py's N > 1 control
flow - process group, id broadcast, iteration sharding, warm-up / timed region, barrier, max over ranks, film reduce, the JSON line -
runs line by line under gloo (tests/test_multi_gpu_gloo.py). An "iteration" is a deterministic image that depends on its index only;
a reduce all-reduces a COPY of the sums and of the iteration count like etx_hip_reduce_film* does over RCCL: out of place (the context's own
sums stay its own), not terminal (rendering continues), any number of times per run - every rank the same number."""
import numpy as np

from tests.lazy_torch import torch, dist

from etx_tracer_amd import api


def fake_iteration(iteration, h=12, w=16):
    g = torch.Generator().manual_seed(1234 + iteration)
    camera = torch.rand((h, w, 4), generator=g) * (1.0 + 0.01 * iteration)
    light = torch.rand((h, w, 4), generator=g) * 0.1
    return camera, light


def pixel_shard_images(camera, light, iteration, pixel_first, pixel_stride):
    """What a pixel-sharded context adds for one iteration (etx_hip_begin_ex): the camera values of ITS pixels (zero elsewhere) and the
    splats of its own light paths anywhere on the frame - here a deterministic partition of the light image whose parts sum to it."""
    if pixel_stride == 1:
        return camera, light
    h, w = camera.shape[:2]
    owned = (torch.arange(h * w).reshape(h, w, 1) % pixel_stride) == pixel_first
    g = torch.Generator().manual_seed(99 + iteration)
    parts = torch.rand((pixel_stride, h, w, 1), generator=g)
    parts = parts / parts.sum(dim=0, keepdim=True)
    return camera * owned, light * parts[pixel_first]


class StubContext:
    instances = []

    def __init__(self, device=0):
        self.device = device
        self.library = None
        self.calls = []
        self.film_size = (0, 0)
        self.iterations_rendered = []  # since the last begin
        self.camera_sum = self.light_sum = None
        self.count = torch.zeros((1,), dtype=torch.int64)
        self.reduces = 0            # since comm_init
        self.reduce_log = []        # per reduce of the current run: iterations this rank had rendered when its snapshot was taken
        self.reduced_camera = self.reduced_light = self.reduced_count = None
        StubContext.instances.append(self)

    def make_unique_id(self):  # multi_gpu.init_context_comm: instead of etx_hip_comm_unique_id
        return bytes(range(128))

    def comm_init(self, rank, world, unique_id):
        self.calls.append(("comm_init", rank, world, bytes(unique_id)))

    def set_debug_flags(self, flags):
        self.calls.append(("set_debug_flags", flags))

    def set_bvh_builder(self, builder):
        self.calls.append(("set_bvh_builder", builder))

    def upload_scene(self, snapshot):
        self.film_size = snapshot.film_size
        self.calls.append(("upload_scene",))

    def bvh_info(self):
        return {"nodes": 1, "triangles": 32, "depth": 1, "stack_need": 3, "bytes": 128, "build_ms": 0.0}

    def upload_cie_table(self, xyz, first):
        self.calls.append(("upload_cie_table",))

    def upload_bluenoise(self, set_index, table):
        self.calls.append(("upload_bluenoise", set_index))

    def device_bytes(self):
        return 0

    def lanes(self, integrator=1):
        return 4

    def set_timers(self, mask):
        self.calls.append(("set_timers", mask))

    def _begin(self, first_iteration, iteration_stride, pixel_first=0, pixel_stride=1):
        self.next_iteration, self.stride = first_iteration, iteration_stride
        self.pixel_first, self.pixel_stride = pixel_first, pixel_stride
        self.iterations_rendered = []
        self.camera_sum = torch.zeros((12, 16, 4))
        self.light_sum = torch.zeros((12, 16, 4))
        self.count.zero_()
        self.reduce_log = []
        self.reduced_camera = self.reduced_light = self.reduced_count = None

    def begin_vcm(self, options, first_iteration=0, iteration_stride=1):
        self.calls.append(("begin_vcm", first_iteration, iteration_stride))
        self._begin(first_iteration, iteration_stride)

    def begin_bdpt(self, options, first_iteration=0, iteration_stride=1, pixel_first=0, pixel_stride=1):
        self.calls.append(("begin_bdpt", first_iteration, iteration_stride, pixel_first, pixel_stride))
        self._begin(first_iteration, iteration_stride, pixel_first, pixel_stride)

    def render_iteration(self):
        camera, light = pixel_shard_images(*fake_iteration(self.next_iteration), self.next_iteration, self.pixel_first, self.pixel_stride)
        self.camera_sum += camera
        self.light_sum += light
        self.count += 1
        self.iterations_rendered.append(self.next_iteration)
        self.next_iteration += self.stride

    def sync(self):
        pass

    def stats(self):
        s = api.Stats()
        n = len(self.iterations_rendered)
        s.completed_iterations = n
        s.rays_extension, s.rays_shadow, s.light_vertices = 1000 * n, 2000 * n, 300 * n
        s.ms_trace_closest, s.launches_trace_closest = 0.5 * n, 10 * n
        s.wavefront_bounces = 20 * n
        return s

    def _reduce(self):
        camera, light = self.camera_sum.clone(), self.light_sum.clone()  # the snapshot: the context's own sums are never modified
        counted = self.count.clone() if self.pixel_first == 0 else torch.zeros_like(self.count)  # pixel shards hold the SAME iterations: the job's iteration counter is shard 0's (per pixel, the device reduces each owner's count)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(camera, op=dist.ReduceOp.SUM)
            dist.all_reduce(light, op=dist.ReduceOp.SUM)
            dist.all_reduce(counted, op=dist.ReduceOp.SUM)
        self.reduced_camera, self.reduced_light, self.reduced_count = camera, light, counted
        self.reduces += 1
        self.reduce_log.append(len(self.iterations_rendered))
        # what the LAST reduce held (bench.py renders more afterwards: its per-kernel pass on rank 0)
        self.reduced_result = self.result()
        self.reduced_iterations = list(self.iterations_rendered)

    def reduce_film(self):
        self.calls.append(("reduce_film",))
        self._reduce()

    def reduce_film_begin(self):
        self.calls.append(("reduce_film_begin",))
        self._reduce()

    def reduce_film_end(self, wait=True):
        return True

    def reduce_info(self):
        info = api.ReduceInfo()
        info.reduces = self.reduces
        info.payload_bytes = int(self.camera_sum.numel() + self.light_sum.numel()) * 4 if self.camera_sum is not None else 0
        info.global_iterations = int(self.reduced_count.item()) if self.reduced_count is not None else 0
        info.last_device_ms, info.total_device_ms = 0.25, 0.25 * self.reduces
        info.layer_mask = 3
        return info

    def result(self):
        """The whole-job image of the newest reduce (etx_hip_read_film on a context with a communicator)."""
        out = torch.clamp((self.reduced_camera + self.reduced_light) / max(int(self.reduced_count.item()), 1), min=0.0)
        out[..., 3] = 1.0
        return out.numpy()

    def read_film(self, layer):
        w, h = self.film_size
        return np.zeros((h, w, 4), dtype=np.float32)

    def trace_rays_device(self, d_o, d_d, count, d_hits, repeat):
        return 0.05

    def trace_rays_timed(self, rays, repeat, want_hits=False):
        return 0.05

    def comm_all_reduce(self, values, op=api.REDUCE_SUM):
        """etx_hip_comm_all_reduce_f64 over the test's gloo group"""
        self.calls.append(("comm_all_reduce", int(op)))
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(t, op={api.REDUCE_SUM: dist.ReduceOp.SUM, api.REDUCE_MAX: dist.ReduceOp.MAX, api.REDUCE_MIN: dist.ReduceOp.MIN}[int(op)])
        return [float(v) for v in t]

    def comm_barrier(self):
        self.calls.append(("comm_barrier",))
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()

    def close(self):
        self.calls.append(("close",))
