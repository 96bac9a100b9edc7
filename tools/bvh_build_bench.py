"""Tree build and traversal cost of the two builders (etx_hip_set_bvh_builder) on scenes of 10^4..10^6 triangles: the gems scene with
scaled copies of its gems scattered through the box (tools/synthetic_scenes.py replicate_gems). Per scene and builder: build time
(etx_hip_bvh_info), nodes / depth / stack bound, closest-hit throughput of 2 M incoherent rays with device-resident queues
(etx_hip_trace_rays_timed: uploaded once, HIP events; no torch in the process), and for the host tree the time of an in-place refit (etx_hip_update_scene, wall clock incl.
the vertex copy). One JSON line per row. Usage: python tools/bvh_build_bench.py [copies ...]
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
etx = importlib.import_module("etx-tracer_amd")
from tools.synthetic_scenes import replicate_gems  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
N_RAYS = 1 << 21


def rays():
    g = np.random.default_rng(1)
    r = np.empty((N_RAYS, 8), dtype=np.float32)
    r[:, 0] = g.random(N_RAYS, dtype=np.float32) * 1.9 - 0.95
    r[:, 1] = g.random(N_RAYS, dtype=np.float32) * 1.85 + 0.05
    r[:, 2] = g.random(N_RAYS, dtype=np.float32) * 1.9 - 0.95
    d = g.standard_normal((N_RAYS, 3), dtype=np.float32)
    r[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 3] = 2.2889e-4
    r[:, 7] = 3.0e38
    return r


def main():
    queue = rays()
    for copies in [int(a) for a in sys.argv[1:]] or [0, 40, 350]:
        snap = replicate_gems(etx, os.path.join(GOLDEN, "cornell_gems_128.etxscene"), copies) if copies else etx.SceneSnapshot(os.path.join(GOLDEN, "cornell_gems_128.etxscene"))
        reference = None
        for name, builder in (("host binned SAH", etx.api.BVH_HOST_SAH), ("device linear", etx.api.BVH_DEVICE_LBVH)):
            ctx = etx.api.Context(0)
            ctx.set_bvh_builder(builder)
            t0 = time.perf_counter()
            ctx.upload_scene(snap)
            upload_s = time.perf_counter() - t0
            info = ctx.bvh_info()
            ms, hits = ctx.trace_rays_timed(queue, 20, want_hits=True)
            found = hits[:, 3].view(np.int32)
            row = {"triangles": int(snap.triangle_count), "builder": name, "build_ms": round(info["build_ms"], 3), "upload_s": round(upload_s, 3), "nodes": info["nodes"],
                   "depth": info["depth"], "stack_need": info["stack_need"], "trace_ms_2M_rays": round(ms, 4), "grays_per_s": round(N_RAYS / ms / 1.0e6, 3),
                   "hit_fraction": round(float((found != -1).mean()), 4)}
            if reference is None:
                reference = found.copy()
                vertices = snap.vertices()
                moved = vertices[:, 0:3].copy()
                vertices[:, 1] += np.float32(1.0e-3)  # every vertex moves: a full refit
                t0 = time.perf_counter()
                ctx.update_scene(snap, etx.api.CHANGED_POSITIONS)
                row["refit_s_wall"] = round(time.perf_counter() - t0, 4)
                t0 = time.perf_counter()
                ctx.update_scene(snap, etx.api.CHANGED_POSITIONS | etx.api.REBUILD_BVH)
                row["device_rebuild_s_wall"] = round(time.perf_counter() - t0, 4)
                row["device_rebuild_ms"] = round(ctx.bvh_info()["build_ms"], 3)
                vertices[:, 0:3] = moved
            else:
                row["same_hits_as_sah_tree"] = round(float((found == reference).mean()), 6)
            print(json.dumps(row), flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
