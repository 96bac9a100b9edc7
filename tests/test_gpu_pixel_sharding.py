"""GPU (-m gpu): pixel-interleaved sharding of the bidirectional integrator and the path tracer (etx_hip_begin_ex).

Two contexts on the one device a test box has render pixels 0, 2, 4, ... and 1, 3, 5, ... of every iteration. Paths are seeded by
(pixel, iteration), so the two contexts trace exactly the paths of the unsharded render: the camera image of each is tile-local (zero on the
other's pixels), the light image full-frame (bidirectional.cxx:516-520: splats land anywhere), and the SUM of the two films is the
unsharded film within fp32 summation order (float atomics). Each context starts with half the growable pools.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def render(etx, golden_dir, cls, flavour, spp, options, pixel_first=0, pixel_stride=1):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_%s_128.etxscene" % flavour))
    snap.samples = spp
    snap.noise_threshold = 0.0
    integ = cls(snap, pixel_first=pixel_first, pixel_stride=pixel_stride)
    integ.options().update(options)
    integ.render()
    cam, light = integ.film(etx.api.LAYER_CAMERA), integ.film(etx.api.LAYER_LIGHT)
    stats = integ.status()
    pool_bytes = integ.context.device_bytes()
    integ.context.close()
    assert stats.completed_iterations == spp and stats.overflow_flags == 0
    assert np.isfinite(cam).all() and np.isfinite(light).all()
    return cam[..., :3], light[..., :3], stats, pool_bytes


@pytest.mark.parametrize("flavour", ["classic", "full", "glass"])
def test_bdpt_two_pixel_shards_sum_to_the_unsharded_film(etx, golden_dir, flavour):
    """glass: rough dielectric / conductor / plastic BSDFs evaluated stochastically - the stream of a connection is keyed by its two vertices' own seeds
    (kernels_bdpt.hip bdpt_connect_pair), not by a pool slot, so a shard evaluates exactly what the unsharded render evaluates."""
    options = {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False}
    spp = 16
    cam, light, whole, whole_bytes = render(etx, golden_dir, etx.HIPBidirectional, flavour, spp, options)
    parts = [render(etx, golden_dir, etx.HIPBidirectional, flavour, spp, options, first, 2) for first in (0, 1)]
    h, w = cam.shape[:2]
    owner = (np.arange(h * w).reshape(h, w) % 2)[::-1]  # film rows are stored bottom-up (film.cxx:189): pixel id = x + (H - 1 - row) * W
    for first, (part_cam, part_light, stats, part_bytes) in enumerate(parts):
        assert float(np.abs(part_cam[owner != first]).max()) == 0.0  # tile-local camera image
        assert float(part_cam[owner == first].sum()) > 0.0
        assert float(part_light.sum()) > 0.0                         # the splats of its own light paths, anywhere on the frame
        assert stats.rays_extension < 0.6 * whole.rays_extension     # half the paths
        assert part_bytes < 0.75 * whole_bytes                       # half the vertex / pair / queue pools (path sets and film stay full-frame)
    assert parts[0][2].rays_extension + parts[1][2].rays_extension == whole.rays_extension  # exactly the unsharded render's paths
    np.testing.assert_allclose(parts[0][0] + parts[1][0], cam, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(parts[0][1] + parts[1][1], light, rtol=2e-4, atol=2e-5)


def test_path_tracer_three_pixel_shards(etx, golden_dir):
    options = {"bn": False}
    cam, light, whole, _ = render(etx, golden_dir, etx.HIPPathTracing, "classic", 8, options)
    parts = [render(etx, golden_dir, etx.HIPPathTracing, "classic", 8, options, first, 3) for first in (0, 1, 2)]
    assert sum(p[2].rays_extension for p in parts) == whole.rays_extension
    np.testing.assert_allclose(sum(p[0] for p in parts), cam, rtol=2e-4, atol=2e-5)


def test_vcm_refuses_pixel_sharding(etx, golden_dir):
    from etx_tracer_amd import api
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_classic_128.etxscene"))
    integ = etx.HIPVCM(snap, pixel_first=0, pixel_stride=2)
    with pytest.raises(api.EtxHipError, match="photon map"):
        integ.render()
    integ.context.close()


def test_vcm_render_is_reproducible_on_stochastic_bsdfs(etx, golden_dir):
    """Two renders of the rough-material box: every random stream - also those of the stochastically evaluated BSDFs in connections and merges and
    the alpha / medium streams of the ray queries - is keyed by path data (pixel, iteration, index in path, the photon's own values), never by a pool
    or queue position, so the films differ by fp32 summation order only (float atomics, several iterations in flight)."""
    films = []
    for _ in range(2):
        snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_rough_128.etxscene"))
        snap.samples = 16
        integ = etx.HIPVCM(snap)
        integ.options()["vcm-blue_noise"] = False
        integ.render()
        films.append((integ.film(etx.api.LAYER_CAMERA)[..., :3], integ.film(etx.api.LAYER_LIGHT)[..., :3]))
        integ.context.close()
    for a, b in zip(*films):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)
