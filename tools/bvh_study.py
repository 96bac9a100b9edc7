#!/usr/bin/env python3
"""What a ray costs in the node formats of tools/bvh_study_src/bvh_study.cpp, on the host (no GPU) - a design study, not part of the product
library (the tool builds its own small library on first use and links it against libetx_hip.so for the host tree builder):

    python3 tools/bvh_study.py [gems] [dragon] [gems1m] [--rays 200000]

Scenes: the 2 892-triangle gems box, the 102 400-triangle subsurface meshes of configs[3] (tools/synthetic_scenes.py sss_dragon), the gems
box grown to a million triangles (replicate_gems). Rays: incoherent (uniform origins in the box, uniform directions), as in the traversal
benches. Per format: nodes and their bytes, node visits / triangle tests per ray, node bytes fetched per ray, the deepest stack.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import etx_tracer_amd as etx  # noqa: E402
from etx_tracer_amd import api  # noqa: E402
from tools import synthetic_scenes  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
STUDY_SOURCE = os.path.join(ROOT, "tools", "bvh_study_src", "bvh_study.cpp")
STUDY_LIBRARY = os.path.join(ROOT, "tools", "bvh_study_src", "libetx_bvh_study.so")
_study = None


def study_library():
    """tools/bvh_study_src/libetx_bvh_study.so, compiled on first use (hipcc, host code only)."""
    global _study
    if _study is not None:
        return _study
    import ctypes
    import subprocess
    product = api.library_path()
    stale = (not os.path.exists(STUDY_LIBRARY)) or (os.path.getmtime(STUDY_LIBRARY) < max(os.path.getmtime(STUDY_SOURCE), os.path.getmtime(product)))
    if stale:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", STUDY_SOURCE, "-o", STUDY_LIBRARY,
                               "-L" + os.path.dirname(product), "-l:" + os.path.basename(product), "-Wl,-rpath," + os.path.dirname(product)], stderr=subprocess.DEVNULL)
    api.Library.get()  # the product library first: the study resolves the host tree builder in it
    _study = ctypes.CDLL(STUDY_LIBRARY)
    _study.etx_bvh_study.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64 * 8), ctypes.c_void_p]
    return _study


def host_bvh_study(snapshot, rays, width=4, quantised=False, sorted_pushes=True, with_hits=False):
    """What `rays` cost in a tree of `width` children per node, boxes exact or 8-bit (etx_bvh_study)."""
    import ctypes
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = (ctypes.c_uint64 * 8)()
    hits = np.zeros((rays.shape[0], 2), dtype=np.float32) if with_hits else None
    rc = study_library().etx_bvh_study(snapshot.scene_address, int(width), int(bool(quantised)), int(bool(sorted_pushes)), rays.ctypes.data, rays.shape[0], ctypes.byref(out),
                                       hits.ctypes.data if with_hits else None)
    result = {"node_visits": out[0], "triangle_tests": out[1], "hits": out[2], "max_stack": out[3], "nodes": out[4], "levels": out[5], "visits_to_final_hit": out[6],
              "max_visits_of_a_ray": out[7]}
    if with_hits:
        triangle = hits[:, 1].view(np.uint32).astype(np.int64)
        triangle[triangle == 0xFFFFFFFF] = -1
        result["t"], result["triangle"] = hits[:, 0].copy(), triangle
    return rc, result


def make_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = np.stack([rng.uniform(-0.95, 0.95, n), rng.uniform(0.05, 1.9, n), rng.uniform(-0.95, 3.5, n)], axis=1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 2.2889e-4, d, 3.4e38
    return rays


def walk_rays(snap, n, seed, mean_free_path=0.05):
    """Segments of a subsurface walk: origins just below the surface of the scene's most finely tessellated material (the subsurface
    meshes), uniform directions, exponentially distributed lengths - most end inside the object without reaching its surface."""
    rng = np.random.default_rng(seed)
    tris, verts = snap.triangles(), snap.vertices()
    material = np.bincount(tris[:, 3]).argmax()
    mesh = tris[tris[:, 3] == material]
    pick = mesh[rng.integers(0, len(mesh), n)]
    p0, p1, p2 = verts[pick[:, 0], 0:3], verts[pick[:, 1], 0:3], verts[pick[:, 2], 0:3]
    centre = (p0 + p1 + p2) / 3.0
    normal = pick[:, 4:7].copy().view(np.float32)
    o = centre - normal * rng.exponential(mean_free_path, (n, 1))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 2.2889e-4, d, rng.exponential(mean_free_path, n)
    return rays


def scene(name):
    if name == "gems":
        return etx.SceneSnapshot(os.path.join(GOLDEN, "cornell_gems_128.etxscene"))
    if name == "dragon":
        return synthetic_scenes.sss_dragon(etx, os.path.join(GOLDEN, "cornell_sss_1080p.etxscene"))
    if name == "gems1m":
        return synthetic_scenes.replicate_gems(etx, os.path.join(GOLDEN, "cornell_gems_128.etxscene"), 350)
    raise SystemExit("unknown scene " + name)


FORMATS = [  # label, width, quantised, sorted pushes, bytes per node
    ("BVH4 float, 128 B (today)", 4, False, True, 128),
    ("BVH4 8-bit, 64 B", 4, True, True, 64),
    ("BVH8 float, sorted", 8, False, True, 256),
    ("BVH8 8-bit, 128 B, sorted", 8, True, True, 128),
    ("BVH8 8-bit, 128 B, nearest first only", 8, True, False, 128),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["gems", "dragon", "gems1m"])
    ap.add_argument("--rays", type=int, default=200000)
    ap.add_argument("--walk", action="store_true", help="segments of a subsurface walk instead of incoherent rays through the box")
    args = ap.parse_args()
    for name in args.scenes:
        snap = scene(name)
        rays = walk_rays(snap, args.rays, 31) if args.walk else make_rays(args.rays, 31)
        print("%s: %d triangles, %d %s" % (name, snap.triangle_count, args.rays, "walk segments (origins below the mesh surface, exponential lengths, mean 0.05)" if args.walk else "incoherent rays"))
        print("  %-40s %9s %9s %7s %12s %12s %12s %9s %9s" % ("format", "nodes", "MB", "levels", "visits/ray", "tests/ray", "node B/ray", "max stack", "max visits"))
        base = None
        for label, width, quantised, sorted_pushes, node_bytes in FORMATS:
            rc, w = host_bvh_study(snap, rays, width=width, quantised=quantised, sorted_pushes=sorted_pushes, with_hits=True)
            assert rc == 0, label
            if base is None:
                base = w
            else:
                same = w["triangle"] == base["triangle"]
                assert (w["hits"] == base["hits"]) and (same.mean() > 0.9995), label
            print("  %-40s %9d %9.2f %7d %12.2f %12.2f %12.0f %9d %9d" % (label, w["nodes"], w["nodes"] * node_bytes / 1.0e6, w["levels"], w["node_visits"] / args.rays,
                                                                          w["triangle_tests"] / args.rays, w["node_visits"] * node_bytes / args.rays, w["max_stack"], w["max_visits_of_a_ray"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
