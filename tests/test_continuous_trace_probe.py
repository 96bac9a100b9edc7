"""Raytracing::continuous_trace beyond its buffer (missing 5).

The Christensen-Burley gather shoots three probe rays and keeps at most eight hits of the object's material along each
(kIntersectionsPerDirection, scene_bssrdf_subsurface.hxx:5). The reference keeps the FIRST eight candidates its traversal accepts
(rt.cxx:412-421: the filter callback appends until the buffer is full) - which eight that is depends on the order Embree visits
candidates in. The device finds the hits one after the other with a material-filtered closest-hit query (dev_sss.h sss_gather_cb): the
eight NEAREST. Scene: the subsurface box cut into six separate slabs (tools/synthetic_scenes.py sss_sheets) - a probe ray along the
stack meets twelve surfaces of one material.

CPU (live oracle/_ref/etx_oracle --probe-continuous-trace, the reference's own function over the oracle's BVH shim): near-first
traversal keeps the eight nearest, far-first the eight farthest, a random child order a mixture - beyond eight hits the reference's
result is a property of its traversal order, as its film is (tests/test_reference_order_spread.py). GPU: the device's eight hits are
the near-first list, hit for hit.
"""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")


def slab_scene(etx, golden_dir, tmp_path):
    from tools import synthetic_scenes
    snap = synthetic_scenes.sss_sheets(etx, os.path.join(golden_dir, "cornell_ssscb_128.etxscene"))
    material = int(np.nonzero(snap.materials()[:, 26] != 0)[0][0])
    tris, verts = snap.triangles(), snap.vertices()
    corners = verts[tris[tris[:, 3] == material, 0:3].reshape(-1).astype(np.int64), 0:3]
    lo, hi = corners.min(axis=0), corners.max(axis=0)
    origin = np.array([(lo[0] + hi[0]) / 2 + 0.013, hi[1] + 0.05, (lo[2] + hi[2]) / 2 + 0.007], dtype=np.float32)
    path = str(tmp_path / "sheets.etxscene")
    snap.save(path)
    return snap, path, material, origin


def reference_hits(path, origin, material, order):
    out = subprocess.run([ORACLE, "--load-snapshot", path, "--integrator", "none", "--probe-continuous-trace", *[repr(float(v)) for v in origin], "0", "-1", "0", "10",
                          str(material), "8"], capture_output=True, text=True, env=dict(os.environ, ETX_ORACLE_BVH_ORDER=order)).stdout
    return json.loads(re.search(r"PROBE_HITS (\{.*\})", out).group(1))


def brute_force_hits(snap, origin, material):
    """every crossing of the ray (origin, -y) with a triangle of `material`, sorted by t (numpy Moeller-Trumbore)"""
    tris, verts = snap.triangles(), snap.vertices()
    mine = np.nonzero(tris[:, 3] == material)[0]
    p0, p1, p2 = (verts[tris[mine, k].astype(np.int64), 0:3].astype(np.float64) for k in range(3))
    d = np.array([0.0, -1.0, 0.0])
    e1, e2 = p1 - p0, p2 - p0
    pv = np.cross(d, e2)
    det = (e1 * pv).sum(axis=1)
    ok = np.abs(det) > 1e-12
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tv = origin.astype(np.float64) - p0
    u = (tv * pv).sum(axis=1) * inv
    qv = np.cross(tv, e1)
    v = (qv * d).sum(axis=1) * inv
    t = (e2 * qv).sum(axis=1) * inv
    hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 1e-4)
    order = np.argsort(t[hit])
    return t[hit][order], mine[hit][order]


@pytest.mark.skipif(not os.path.exists(ORACLE), reason="oracle/_ref/etx_oracle is not built here")
def test_reference_keeps_the_first_eight_in_traversal_order(etx, golden_dir, tmp_path):
    snap, path, material, origin = slab_scene(etx, golden_dir, tmp_path)
    t_all, tri_all = brute_force_hits(snap, origin, material)
    assert len(t_all) == 12  # six slabs, two faces each
    near, far, rnd = (reference_hits(path, origin, material, order) for order in ("near_first", "far_first", "random_child"))
    assert near["count"] == far["count"] == rnd["count"] == 8
    np.testing.assert_allclose(sorted(near["t"]), t_all[:8], rtol=1e-5)
    assert sorted(near["triangle"]) == sorted(int(i) for i in tri_all[:8])
    np.testing.assert_allclose(sorted(far["t"]), t_all[4:], rtol=1e-5)   # the eight FARTHEST
    assert set(rnd["triangle"]) <= set(int(i) for i in tri_all) and set(rnd["triangle"]) != set(near["triangle"]) and set(rnd["triangle"]) != set(far["triangle"])


@pytest.mark.gpu
def test_device_keeps_the_eight_nearest(etx, golden_dir, tmp_path):
    """sss_gather_cb's loop restated through the C ABI: closest hit, step behind it (t_min = t + max(eps, t * 1e-6), dev_sss.h), again - eight times.
    (The slabs are the only geometry the ray meets before the floor, so the query needs no material filter.)"""
    from etx_tracer_amd import api
    snap, path, material, origin = slab_scene(etx, golden_dir, tmp_path)
    ctx = api.Context(0)
    ctx.upload_scene(snap)
    t_min, found_t, found_tri = 2.2889e-4, [], []
    for _ in range(8):
        ray = np.array([[origin[0], origin[1], origin[2], t_min, 0.0, -1.0, 0.0, 10.0]], dtype=np.float32)
        hit = ctx.trace_rays(ray)[0]
        triangle = int(hit[3:4].view(np.uint32)[0])
        assert triangle != 0xFFFFFFFF
        found_t.append(float(hit[2])), found_tri.append(triangle)
        t_min = float(hit[2]) + max(2.2889e-4, float(hit[2]) * 1.0e-6)
    ctx.close()
    t_all, tri_all = brute_force_hits(snap, origin, material)
    np.testing.assert_allclose(found_t, t_all[:8], rtol=1e-5)
    assert found_tri == [int(i) for i in tri_all[:8]]
    if os.path.exists(ORACLE):  # = the reference's list under near-first traversal, hit for hit
        near = reference_hits(path, origin, material, "near_first")
        assert sorted(near["triangle"]) == sorted(found_tri)
