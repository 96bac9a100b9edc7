"""CPU: the eight-wide tree with 8-bit child boxes that ETX_HIP_BVH_WIDE uploads (host_scene.cpp encode_bvh8 -> dev_scene.h Bvh8Node),
walked on the host through the node function the kernels call (dev_bvh8.h bvh8_visit: byte decoding, the ray moved into the node's
frame, the folded slab test). It must find the closest hits of today's four-wide tree ray by ray - incoherent rays, short rays,
axis-parallel rays (a zero direction component makes the folded test drop that axis, which may only add visits), rays that start on
a box face - and, as an occlusion walk, report a blocker exactly for the rays that have a hit."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import make_rays
from tools import bvh_study


def rays_for(seed, n=30000):
    rays = make_rays(n, seed)
    rays[:300, 7] = 0.4                                   # short segments
    rays[300:600, 4:7] = np.array([0.0, 0.0, 1.0])        # axis-parallel: two zero components
    rays[600:900, 4:7] = np.array([1.0, 0.0, 0.0])
    rays[900:1200, 4] = 0.0                               # one zero component
    rays[900:1200, 4:7] /= np.linalg.norm(rays[900:1200, 4:7], axis=1, keepdims=True)
    rays[1200:1500, 0:3] = np.round(rays[1200:1500, 0:3] * 8.0) / 8.0  # origins on round coordinates (box faces of the Cornell geometry)
    return rays


@pytest.mark.parametrize("scene", ["cornell_gems_128", "cornell_sssmesh_128"])
def test_encoded_wide_tree_finds_the_hits_of_the_four_wide_tree(etx, golden_dir, scene):
    from etx_tracer_amd import api
    snap = etx.SceneSnapshot(os.path.join(golden_dir, scene + ".etxscene"))
    rays = rays_for(5)
    rc, today = api.host_bvh_stats(snap, rays, with_hits=True)
    assert rc == 0 and today["hits"] > 5000
    rc, wide = api.host_bvh8_stats(snap, rays, with_hits=True)
    assert rc == 0
    assert wide["hits"] == today["hits"]
    same = wide["triangle"] == today["triangle"]
    assert same.mean() > 0.9995, same.mean()  # ties between coplanar facets aside
    np.testing.assert_array_equal(wide["t"][same], today["t"][same])
    assert wide["node_visits"] < 0.8 * today["node_visits"], (wide["node_visits"], today["node_visits"])
    # the tree's exact stack bound (seven pushes per level at worst) against what rays use; the device accepts bounds up to kMaxWideStackDepth = 128
    assert wide["max_stack"] <= wide["stack_need"] <= 128 and wide["max_stack"] <= 24, (wide["max_stack"], wide["stack_need"])
    # the study's eight-wide format (greedy collapse, exact decode, no margin) against the encoded one (cost-driven collapse with merged
    # leaves): about the same visits, a third to a half of the nodes
    rc, study = bvh_study.host_bvh_study(snap, rays, width=8, quantised=True, sorted_pushes=False)
    assert rc == 0 and wide["nodes"] < 0.6 * study["nodes"] and wide["levels"] <= study["levels"]
    assert abs(wide["node_visits"] / study["node_visits"] - 1.0) < 0.15
    # occlusion walk: a blocker for exactly the rays with a hit (any blocker: the first accepted triangle in traversal order)
    rc, shadow = api.host_bvh8_stats(snap, rays, occlusion=True, with_hits=True)
    assert rc == 0 and shadow["hits"] == today["hits"]
    assert ((shadow["triangle"] >= 0) == (today["triangle"] >= 0)).all()
    assert shadow["node_visits"] <= wide["node_visits"]


def test_encoded_wide_tree_on_a_large_scene(etx, golden_dir):
    """102 400-triangle meshes (configs[3]): walk segments below the surface, the rays the subsurface kernels trace."""
    from etx_tracer_amd import api
    from tools import synthetic_scenes, bvh_study
    snap = synthetic_scenes.sss_dragon(etx, os.path.join(golden_dir, "cornell_sss_1080p.etxscene"))
    rays = np.concatenate([bvh_study.walk_rays(snap, 20000, 3), make_rays(20000, 4)])
    rc, today = api.host_bvh_stats(snap, rays, with_hits=True)
    rc8, wide = api.host_bvh8_stats(snap, rays, with_hits=True)
    assert rc == 0 and rc8 == 0 and wide["hits"] == today["hits"] and today["hits"] > 5000
    same = wide["triangle"] == today["triangle"]
    assert same.mean() > 0.9995
    np.testing.assert_array_equal(wide["t"][same], today["t"][same])
    assert wide["node_visits"] < 0.75 * today["node_visits"]
    assert wide["max_stack"] <= wide["stack_need"] <= 128


@pytest.mark.parametrize("distance", [30.0, 300.0, 3000.0])
def test_encoded_wide_tree_from_far_outside_the_scene(etx, golden_dir, distance):
    """Ray origins 10 - 1000 scene radii away (a distant camera): the folded slab test t = q * step + base loses the low bits of `base`, which grows
    with the distance of the origin from the node - bvh8_ray_frame gives a few ulps of |base| back to the interval. No hit of
    the four-wide tree may be missed."""
    from etx_tracer_amd import api
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_gems_128.etxscene"))
    rng = np.random.default_rng(11)
    n = 20000
    direction = rng.normal(size=(n, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    target = np.stack([rng.uniform(-0.9, 0.9, n), rng.uniform(0.1, 1.9, n), rng.uniform(-0.9, 0.9, n)], axis=1)  # points inside the box
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3] = target - direction * distance
    rays[:, 3] = 2.2889e-4
    rays[:, 4:7] = direction
    rays[:, 7] = 3.4e38
    rc, today = api.host_bvh_stats(snap, rays, with_hits=True)
    rc8, wide = api.host_bvh8_stats(snap, rays, with_hits=True)
    assert rc == 0 and rc8 == 0 and today["hits"] > 0.9 * n
    missed = (today["triangle"] >= 0) & (wide["triangle"] < 0)
    assert missed.sum() == 0, missed.sum()
    same = wide["triangle"] == today["triangle"]
    assert same.mean() > 0.999, same.mean()  # at this distance neighbouring facets tie within the float spacing of t
    assert wide["node_visits"] < 1.2 * today["node_visits"]  # the slack admits a few more nodes, not many
