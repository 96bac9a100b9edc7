"""Scenes of 10^4..10^6 triangles without scene files of that size: the gems snapshot with scaled copies of its gem meshes scattered
through the box (real shape statistics: small closed facetted objects). Used by tests/test_gpu_scene_update.py, tests/test_host_lbvh.py,
tools/bvh_build_bench.py and bench.py --workload gems1m."""
import numpy as np


def replicate_gems(etx, snapshot_path, copies, seed=9):
    """-> SceneSnapshot of `snapshot_path` (a cornell_gems_* snapshot) with `copies` extra copies of its conductor / dielectric
    triangles: 2 892 + copies * 2 880 triangles. Emissive triangles keep their indices (emitter instances name them)."""
    snap = etx.SceneSnapshot(snapshot_path)
    vertices, triangles, to_emitter = snap.vertices().copy(), snap.triangles().copy(), snap.triangle_to_emitter().copy()
    classes = snap.material_classes()
    gem = np.nonzero(np.isin(classes[triangles[:, 3]], (3, 4)))[0]
    rng = np.random.default_rng(seed)
    new_vertices, new_triangles = [vertices], [triangles]
    base = vertices.shape[0]
    corner_rows = vertices[triangles[gem, 0:3].reshape(-1).astype(np.int64)]  # (3 * G, 14): unshared copies
    centre = corner_rows[:, 0:3].mean(axis=0)
    for _ in range(copies):
        scale = np.float32(rng.uniform(0.1, 0.3))
        offset = np.float32([rng.uniform(-0.8, 0.8), rng.uniform(0.15, 1.8), rng.uniform(-0.8, 0.8)])
        rows = corner_rows.copy()
        rows[:, 0:3] = (rows[:, 0:3] - centre) * scale + offset
        tri = triangles[gem].copy()
        tri[:, 0:3] = base + np.arange(3 * len(gem), dtype=np.uint32).reshape(-1, 3)
        new_vertices.append(rows)
        new_triangles.append(tri)
        base += rows.shape[0]
    all_triangles = np.concatenate(new_triangles)
    snap.replace_geometry(np.concatenate(new_vertices), all_triangles, np.concatenate([to_emitter, np.full(all_triangles.shape[0] - to_emitter.shape[0], 0xFFFFFFFF, dtype=np.uint32)]))
    return snap
