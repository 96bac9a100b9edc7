"""CPU: the scenes bench.py assembles at run time for BASELINE.json configs[3] and configs[4] (tools/synthetic_scenes.py,
SceneSnapshot.inject_density / save) - what the device AND the reference's driver are handed."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def etx_cpu():
    import sys
    sys.path.insert(0, ROOT)
    import etx_tracer_amd
    return etx_tracer_amd


def test_sss_dragon_is_two_closed_outward_meshes(etx_cpu):
    from tools import synthetic_scenes
    base = etx_cpu.SceneSnapshot(os.path.join(GOLDEN, "cornell_sss_1080p.etxscene"))
    snap = synthetic_scenes.sss_dragon(etx_cpu, os.path.join(GOLDEN, "cornell_sss_1080p.etxscene"))
    assert snap.film_size == (1920, 1080)
    triangles, vertices = snap.triangles(), snap.vertices()
    sss = np.nonzero(snap.materials()[:, 26] != 0)[0]
    assert len(sss) == 2
    counts = [int((triangles[:, 3] == m).sum()) for m in sss]
    assert counts == [20 * 4 ** 6, 20 * 4 ** 5] and snap.triangle_count == sum(counts) + 12  # + room and light
    assert snap.triangle_count >= 100000
    for m in sss:
        t = triangles[triangles[:, 3] == m]
        idx = t[:, 0:3].astype(np.int64)
        # closed 2-manifold: every undirected edge belongs to exactly two triangles, once in each direction
        e = np.concatenate([idx[:, [0, 1]], idx[:, [1, 2]], idx[:, [2, 0]]])
        key = e[:, 0] * (1 << 32) + e[:, 1]
        rev = e[:, 1] * (1 << 32) + e[:, 0]
        assert len(np.unique(key)) == len(key) and np.array_equal(np.sort(key), np.sort(rev))
        # outward: geometric normals point away from the centroid, vertex normals agree with them, the stored geo_n is the cross product's
        p = vertices[:, 0:3].astype(np.float64)
        c = p[np.unique(idx)].mean(axis=0)
        centre = p[idx].mean(axis=1)
        n = np.cross(p[idx[:, 1]] - p[idx[:, 0]], p[idx[:, 2]] - p[idx[:, 0]])
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        assert ((n * (centre - c)).sum(axis=1) > 0).mean() > 0.97  # a knobbly blob: a few faces of a neck look inward of the centroid
        np.testing.assert_allclose(t[:, 4:7].view(np.float32), n, atol=2e-4)
        assert ((vertices[idx[:, 0], 3:6] * n).sum(axis=1) > 0.3).all()
        np.testing.assert_allclose(np.linalg.norm(vertices[np.unique(idx), 3:6], axis=1), 1.0, atol=1e-5)
    # the emitter instances still name emissive triangles
    emitters = snap.emitter_instances()
    area = emitters[emitters[:, 2] != 0xFFFFFFFF, 2]
    assert len(area) > 0 and (snap.triangle_to_emitter()[area] != 0xFFFFFFFF).all()
    assert np.array_equal(snap.triangle_to_emitter()[area], base.triangle_to_emitter()[base.emitter_instances()[base.emitter_instances()[:, 2] != 0xFFFFFFFF, 2]])


def test_snapshot_save_round_trips_replaced_arrays(etx_cpu, tmp_path):
    from tools import synthetic_scenes
    snap = synthetic_scenes.sss_dragon(etx_cpu, os.path.join(GOLDEN, "cornell_sss_1080p.etxscene"), subdivisions=(2, 1))
    snap.samples = 7
    path = str(tmp_path / "assembled.etxscene")
    snap.save(path)
    again = etx_cpu.SceneSnapshot(path)
    assert again.samples == 7 and again.film_size == snap.film_size
    assert np.array_equal(again.triangles(), snap.triangles()) and np.array_equal(again.vertices(), snap.vertices())
    assert np.array_equal(again.triangle_to_emitter(), snap.triangle_to_emitter()) and np.array_equal(again.emitter_instances(), snap.emitter_instances())
    assert np.array_equal(again.materials(), snap.materials())
    # ... and the reference's own driver reads it (where it is built)
    oracle = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")
    if os.path.exists(oracle):
        out = subprocess.run([oracle, "--load-snapshot", path, "--integrator", "none"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert out.returncode == 0 and ("%d triangles" % snap.triangle_count) in out.stdout, out.stdout[-500:]


def test_injected_density_is_the_drivers_grid(etx_cpu, tmp_path):
    """SceneSnapshot.inject_density(n) against the grid `etx_oracle --inject-density n` wrote into the committed 32^3 cloud snapshot:
    the same formula on both sides (1 ulp of float32 libm difference at most), every medium heterogeneous with the grid's dimensions."""
    reference = etx_cpu.SceneSnapshot(os.path.join(GOLDEN, "cornell_cloud_128.etxscene"))
    ptr, count = reference._array(112)  # etx_abi_scene::mediums
    assert count >= 1
    grid_ptr, grid_count = ctypes.c_uint64.from_address(ptr).value, ctypes.c_uint64.from_address(ptr + 8).value
    expected = np.frombuffer((ctypes.c_float * grid_count).from_address(grid_ptr), dtype=np.float32)
    snap = etx_cpu.SceneSnapshot(os.path.join(GOLDEN, "cornell_full_128.etxscene"))
    grid = snap.inject_density(32)
    assert grid.shape == expected.shape == (32 ** 3,)
    assert float(np.abs(grid - expected).max()) <= 2.4e-7 and float(grid.max()) == 1.0
    mptr, mcount = snap._array(112)
    for i in range(mcount):
        m = mptr + 80 * i
        assert ctypes.c_uint16.from_address(m + 48).value == 1 and [ctypes.c_uint32.from_address(m + 68 + 4 * k).value for k in range(3)] == [32, 32, 32]
        assert ctypes.c_uint64.from_address(m + 8).value == 32 ** 3
    path = str(tmp_path / "cloud.etxscene")
    snap.save(path)
    again = etx_cpu.SceneSnapshot(path)
    aptr, _ = again._array(112)
    saved = np.frombuffer((ctypes.c_float * (32 ** 3)).from_address(ctypes.c_uint64.from_address(aptr).value), dtype=np.float32)
    assert np.array_equal(saved, grid)
