"""GPU (-m gpu): the eight-wide tree with 8-bit child boxes (etx_hip_set_bvh_builder(BVH_HOST_SAH | BVH_WIDE), csrc/dev_bvh8.h).

The node function is the one tests/test_host_bvh8.py walks on the host; register budgets in tests/test_build_budget.py. First device
run: round 4 (tools/gpu_calls/gpu_r4a.sh), all cases green.

What they check: the closest hits of the default tree ray by ray (the traversal visits other nodes in another order, a closest hit does
not depend on that), the occlusion-only shadow kernel and the bidirectional walk kernels through renders that must agree with the
default tree's value by value (same seeds, same hits; float addition order aside), and that an edit that moves vertices drops the
wide tree instead of traversing a stale one.
"""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import make_rays
from tests.test_gpu_scene_update import assert_same_render, hit_triangles, render

pytestmark = pytest.mark.gpu


def contexts(etx, snap):
    wide, default = etx.api.Context(0), etx.api.Context(0)
    wide.set_bvh_builder(etx.api.BVH_HOST_SAH | etx.api.BVH_WIDE)
    wide.upload_scene(snap)
    default.upload_scene(snap)
    return wide, default


@pytest.mark.parametrize("scene", ["cornell_gems_128", "cornell_sssmesh_128"])
def test_wide_tree_finds_the_default_trees_hits(etx, golden_dir, scene):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, scene + ".etxscene"))
    wide, default = contexts(etx, snap)
    assert wide.bvh_info()["bytes"] > default.bvh_info()["bytes"]  # both trees are resident
    rays = make_rays(200000, 17)
    rays[:500, 7] = 0.4
    rays[500:1000, 4:7] = np.array([0.0, 0.0, 1.0], dtype=np.float32)  # axis-parallel
    hits_w, hits_d = wide.trace_rays(rays), default.trace_rays(rays)
    same = hit_triangles(hits_w) == hit_triangles(hits_d)
    assert same.mean() > 0.9995 and (hit_triangles(hits_d) >= 0).mean() > 0.3
    np.testing.assert_allclose(hits_w[same, 2], hits_d[same, 2], rtol=1e-6, atol=1e-6)
    wide.close()
    default.close()


def test_wide_tree_on_a_hundred_thousand_triangles(etx, golden_dir):
    from tools import synthetic_scenes, bvh_study
    snap = synthetic_scenes.sss_dragon(etx, os.path.join(golden_dir, "cornell_sss_1080p.etxscene"))
    wide, default = contexts(etx, snap)
    rays = np.concatenate([bvh_study.walk_rays(snap, 100000, 3), make_rays(100000, 4)])
    hits_w, hits_d = wide.trace_rays(rays), default.trace_rays(rays)
    same = hit_triangles(hits_w) == hit_triangles(hits_d)
    assert same.mean() > 0.9995
    np.testing.assert_allclose(hits_w[same, 2], hits_d[same, 2], rtol=1e-6, atol=1e-6)
    wide.close()
    default.close()


@pytest.mark.parametrize("kind", ["bdpt", "vcm", "pt"])
def test_renders_agree_with_the_default_tree(etx, golden_dir, kind):
    """sssmesh: no Boundary material, no density grid - shadow segments take the occlusion-only kernel, the bidirectional integrator's walks
    their own kernels, closest hits the persistent kernel: all three read the wide tree here."""
    films = []
    for builder in (etx.api.BVH_HOST_SAH | etx.api.BVH_WIDE, etx.api.BVH_HOST_SAH):
        snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_sssmesh_128.etxscene"))
        snap.samples = 16
        snap.noise_threshold = 0.0
        cls = {"bdpt": etx.HIPBidirectional, "vcm": etx.HIPVCM, "pt": etx.HIPPathTracing}[kind]
        integ = cls(snap)
        integ.bvh_builder = builder
        integ.options().update({"bdpt": {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False}, "vcm": {"vcm-blue_noise": False}, "pt": {"bn": False}}[kind])
        films.append(render(etx, integ))
        integ.context.close()
    assert_same_render(films[0], films[1], "eight-wide tree vs four-wide tree, " + kind)


def test_moved_vertices_drop_the_wide_tree(etx, golden_dir):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_gems_128.etxscene"))
    ctx = etx.api.Context(0)
    ctx.set_bvh_builder(etx.api.BVH_HOST_SAH | etx.api.BVH_WIDE)
    ctx.upload_scene(snap)
    rays = make_rays(50000, 9)
    before = ctx.trace_rays(rays)
    verts = snap.vertices()
    verts[:, 1] += 0.01  # everything moves up a centimetre
    ctx.update_scene(snap, etx.api.CHANGED_POSITIONS)
    moved = rays.copy()
    moved[:, 1] += 0.01
    after = ctx.trace_rays(moved)  # the refit four-wide tree: the same hits, a centimetre higher
    same = hit_triangles(before) == hit_triangles(after)
    assert same.mean() > 0.999
    np.testing.assert_allclose(after[same, 2], before[same, 2], rtol=1e-4, atol=1e-5)
    verts[:, 1] -= 0.01
    ctx.close()
