"""CPU: the device BVH build (dev_lbvh.h, kernels_bvh_build.hip) emulated on the host - the same per-element functions the kernels
call, run element by element (etx_hip_host_check_bvh_builder / etx_hip_host_bvh_stats_builder with ETX_HIP_BVH_DEVICE_LBVH). What the
device will build is checked here without a GPU: every triangle in exactly one leaf, boxes nested, children numbered after their
parents, the traversal stack bound, and the walk of the device traversal finding the same closest hits as in the binned-SAH tree.
tests/test_gpu_scene_update.py runs the kernels themselves against this."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import make_rays


def load(etx, golden_dir, name="cornell_gems_128"):
    return etx.SceneSnapshot(os.path.join(golden_dir, name + ".etxscene"))


def test_linear_bvh_invariants_and_size(etx, golden_dir):
    from etx_tracer_amd import api
    snap = load(etx, golden_dir)
    rc, info = api.host_check_bvh(snap, builder=api.BVH_DEVICE_LBVH)
    assert rc == 0, info
    assert info["triangles"] == snap.triangle_count
    assert snap.triangle_count / 16 <= info["nodes"] < snap.triangle_count  # four-wide nodes over leaves of up to four triangles
    assert info["stack_need"] <= 32 and info["depth"] <= 16, info
    assert info["bytes"] == info["nodes"] * 128 + info["triangles"] * 48


def test_linear_bvh_finds_the_hits_of_the_sah_tree(etx, golden_dir):
    from etx_tracer_amd import api
    snap = load(etx, golden_dir)
    rays = make_rays(20000, 23)
    rc_s, sah = api.host_bvh_stats(snap, rays, builder=api.BVH_HOST_SAH, with_hits=True)
    rc_l, lbvh = api.host_bvh_stats(snap, rays, builder=api.BVH_DEVICE_LBVH, with_hits=True)
    assert rc_s == 0 and rc_l == 0
    assert sah["hits"] > 5000 and lbvh["hits"] == sah["hits"]
    same = sah["triangle"] == lbvh["triangle"]
    assert same.mean() > 0.9995  # a closest hit does not depend on the tree (ties between coplanar facets aside)
    np.testing.assert_array_equal(lbvh["t"][same], sah["t"][same])
    assert lbvh["max_stack"] <= 32
    # quality: the price of a linear build, in the units the traversal kernel pays
    ratio_nodes = lbvh["node_visits"] / sah["node_visits"]
    ratio_tris = lbvh["triangle_tests"] / sah["triangle_tests"]
    print("linear / SAH: node visits %.2f, triangle tests %.2f" % (ratio_nodes, ratio_tris))
    assert ratio_nodes < 3.0 and ratio_tris < 3.0


def test_linear_bvh_survives_duplicate_keys_and_degenerate_triangles(etx, golden_dir):
    """1280 triangles collapsed onto one point (identical Morton keys: the radix tree splits them by position, Karras 2012 section 4),
    the rest untouched."""
    from etx_tracer_amd import api
    snap = load(etx, golden_dir)
    triangles, vertices = snap.triangles(), snap.vertices()
    counts = np.bincount(triangles[:, 3])
    collapsed = int(np.argmax(counts))
    corners = np.unique(triangles[triangles[:, 3] == collapsed][:, 0:3].reshape(-1))
    vertices[corners, 0:3] = np.float32([0.1, 0.7, 0.2])
    rc, info = api.host_check_bvh(snap, builder=api.BVH_DEVICE_LBVH)
    assert rc == 0 and info["stack_need"] <= 32, info
    rays = make_rays(5000, 4)
    _, sah = api.host_bvh_stats(snap, rays, builder=api.BVH_HOST_SAH, with_hits=True)
    _, lbvh = api.host_bvh_stats(snap, rays, builder=api.BVH_DEVICE_LBVH, with_hits=True)
    assert (sah["triangle"] == lbvh["triangle"]).mean() > 0.9995


def test_small_scenes_keep_the_host_build(etx, golden_dir):
    from etx_tracer_amd import api
    snap = load(etx, golden_dir, "cornell_classic_128")  # 32 triangles: flat sweep, host tables
    rc_a, a = api.host_check_bvh(snap, builder=api.BVH_HOST_SAH)
    rc_b, b = api.host_check_bvh(snap, builder=api.BVH_DEVICE_LBVH)
    assert rc_a == 0 and rc_b == 0
    assert b["triangles"] == a["triangles"] == 32


def test_deep_trees_are_accepted_up_to_the_spill_bound(etx, golden_dir):
    """118 000 triangles: either builder's tree needs more traversal stack than a lane keeps in LDS (32 entries; the bound is three pushed
    children on every level of the deepest path). Such trees are accepted up to kMaxStackDepth = 512 (64 until round 5) - the kernels spill the upper
    part of the stack to global memory (dev_bvh.h LaneStack) - and the walk itself stays far below the bound."""
    from etx_tracer_amd import api
    from tests.test_gpu_scene_update import replicate_gems
    snap = replicate_gems(etx, golden_dir, 40)
    rays = make_rays(20000, 2)
    for builder in (api.BVH_HOST_SAH, api.BVH_DEVICE_LBVH):
        rc, info = api.host_check_bvh(snap, builder=builder)
        assert rc == 0 and 32 < info["stack_need"] <= 64, info
        rc, walk = api.host_bvh_stats(snap, rays, builder=builder)
        assert rc == 0 and walk["max_stack"] < 24, walk


def test_task_parallel_sah_build_is_the_sequential_tree(etx, golden_dir, monkeypatch):
    """Scenes of >= 32 768 triangles are built by tasks (host_scene.cpp Builder::build_range): ranges are disjoint and every split sees
    the primitives in the order the sequential build would have left them, so the tree - and with it every traversal statistic - is
    the sequential one."""
    from etx_tracer_amd import api
    from tests.test_gpu_scene_update import replicate_gems
    snap = replicate_gems(etx, golden_dir, 40)
    rays = make_rays(5000, 6)
    results = []
    for threads in ("1", "8"):
        monkeypatch.setenv("ETX_HIP_BVH_BUILD_THREADS", threads)
        rc, info = api.host_check_bvh(snap)
        assert rc == 0
        rc, walk = api.host_bvh_stats(snap, rays, with_hits=True)
        assert rc == 0
        results.append((info, walk))
    assert results[0][0] == results[1][0]
    for key in ("node_visits", "triangle_tests", "hits", "max_stack"):
        assert results[0][1][key] == results[1][1][key]
    np.testing.assert_array_equal(results[0][1]["triangle"], results[1][1]["triangle"])
    np.testing.assert_array_equal(results[0][1]["t"], results[1][1]["t"])
