"""CPU: the node formats of tools/bvh_study_src/bvh_study.cpp (four / eight children per node, exact or 8-bit outward-rounded boxes, sorted or
unsorted pushes) find the closest hits of the tree the device traverses today (etx_hip_host_bvh_stats: the float BVH4 walked as
dev_bvh.h bvh_closest walks it). A quantised box contains its exact box, so such a walk visits a superset of the nodes; the order of
the pushes changes the work, not the result. tools/bvh_study.py prints the work per format for the bench scenes."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import make_rays
from tools import bvh_study


@pytest.fixture(scope="module")
def gems(etx, golden_dir):
    return etx.SceneSnapshot(os.path.join(golden_dir, "cornell_gems_128.etxscene"))


def test_every_format_finds_the_hits_of_the_device_tree(etx, gems):
    from etx_tracer_amd import api
    rays = make_rays(20000, 41)
    rays[:200, 7] = 0.4  # short rays
    rc, today = api.host_bvh_stats(gems, rays, with_hits=True)
    assert rc == 0 and today["hits"] > 5000
    work = {}
    for width in (4, 8):
        for quantised in (False, True):
            for sorted_pushes in (True, False):
                rc, got = bvh_study.host_bvh_study(gems, rays, width=width, quantised=quantised, sorted_pushes=sorted_pushes, with_hits=True)
                assert rc == 0, (width, quantised, sorted_pushes)
                assert got["hits"] == today["hits"]
                same = got["triangle"] == today["triangle"]
                assert same.mean() > 0.9995, (width, quantised, sorted_pushes, same.mean())  # ties between coplanar facets aside
                np.testing.assert_array_equal(got["t"][same], today["t"][same])
                work[(width, quantised, sorted_pushes)] = got
    # the exact four-wide tree walked with sorted pushes IS today's walk
    assert work[(4, False, True)]["node_visits"] == today["node_visits"] and work[(4, False, True)]["triangle_tests"] == today["triangle_tests"]
    assert work[(4, False, True)]["max_stack"] == today["max_stack"]
    # quantised boxes are supersets: never fewer visits, and not many more
    for width in (4, 8):
        exact, coarse = work[(width, False, True)], work[(width, True, True)]
        assert exact["node_visits"] <= coarse["node_visits"] <= 1.25 * exact["node_visits"]
        assert exact["triangle_tests"] <= coarse["triangle_tests"] <= 1.25 * exact["triangle_tests"]
    # eight children per node: fewer, fatter nodes and fewer levels
    # (gems: 617 against 849 nodes, 5 against 8 levels, 2.0 against 3.2 visits per ray; profiles/round3_bvh_format_study.txt)
    assert work[(8, False, True)]["nodes"] < 0.8 * work[(4, False, True)]["nodes"]
    assert work[(8, False, True)]["levels"] < work[(4, False, True)]["levels"]
    assert work[(8, False, True)]["node_visits"] < 0.75 * work[(4, False, True)]["node_visits"]
    # pushing the other hit children unsorted costs a few per cent of visits
    assert work[(8, True, False)]["node_visits"] < 1.1 * work[(8, True, True)]["node_visits"]


def test_study_rejects_other_widths(etx, gems):
    from etx_tracer_amd import api
    rc, _ = bvh_study.host_bvh_study(gems, make_rays(16, 1), width=6)
    assert rc == -1  # ETX_HIP_ERROR_INVALID_ARGUMENT
