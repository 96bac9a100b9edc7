"""etx-tracer_amd: MI355X (gfx950) wavefront VCM backend for etx-tracer, host-side Python mirror.

The product is `libetx_hip.so` (C ABI in include/etx_hip.h, HIP kernels in csrc/). This package only
  * loads that library (`api`), failing loudly when it is missing - there is no CPU path,
  * relocates scene snapshots written by the reference's own loader (`scene_snapshot`),
  * mirrors the reference's `Integrator` plugin interface for tests and bench.py (`integrator.HIPVCM`, `integrator.HIPPathTracing`, `integrator.HIPBidirectional`),
  * shards iterations over ranks and reduces the film (`multi_gpu`).
"""
from .api import Library, EtxHipError, library_path, VCMOptions, PTOptions, BDPTOptions, Stats  # noqa: F401
from .scene_snapshot import SceneSnapshot  # noqa: F401
from .integrator import HIPVCM, HIPPathTracing, HIPBidirectional, Integrator  # noqa: F401
