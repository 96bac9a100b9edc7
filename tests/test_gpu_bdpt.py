"""GPU (-m gpu): the bidirectional integrator (HIPBidirectional, kernels_bdpt.hip) against the reference's CPUBidirectional.

Golden films: tests/golden/hi/cornell_<flavour>_128_bdpt<mode>_<spp>[_rekeyed].npz (oracle/gen_golden_hi.py --integrators bdpt),
mode = CPUBidirectionalImpl::Mode (0 PathTracing, 1 LightTracing, 2 BDPTFast, 3 BDPTFull), bdpt-blue_noise=false. As for VCM
(test_gpu_parity_hi.py), the reference seeds the light and the camera sub path of a pixel with the same sampler state
(bidirectional.cxx:379-380); the device gives the camera path a stream of its own, so the tight limits are asserted against
the reference run with independent streams (`_rekeyed`, ETX_ORACLE_DECORRELATE=2) and the unmodified reference is compared
with the limits its own correlation leaves. The device renders the same iteration set as two interleaved halves; the
metric (noise-corrected block RMSE, relative mean, bias percentile) is test_gpu_parity_hi.compare.
"""
import os

import numpy as np
import pytest

from tests.test_gpu_parity_hi import compare

pytestmark = pytest.mark.gpu


def load(golden_dir, name):
    path = os.path.join(golden_dir, "hi", name)
    assert os.path.exists(path), "%s is missing: run oracle/gen_golden_hi.py --integrators bdpt in the build container" % path
    return np.load(path)


def render_halves(etx, golden_dir, flavour, spp, options, cie=None, debug_flags=0):
    """The two interleaved halves of the iteration set, two contexts one after the other (as tests/test_gpu_parity_hi.py render_halves)."""
    def half(first):
        snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_%s_128.etxscene" % flavour))
        snap.samples = spp
        integ = etx.HIPBidirectional(snap, first_iteration=first, iteration_stride=2)
        integ.options().update(options)
        integ.cie_table = cie
        if debug_flags:
            integ.context.set_debug_flags(debug_flags)
        integ.render()
        cam, light = integ.film(etx.api.LAYER_CAMERA), integ.film(etx.api.LAYER_LIGHT)
        stats = integ.status()
        integ.context.close()
        return cam, light, stats

    # one after the other: rendering the halves from two host threads at once made the suite 10 % shorter and crashed the interpreter in one of five
    # full runs (GPU call r5k; not reproduced with the fault handler on) - two contexts driven CONCURRENTLY from one process are not something the
    # product promises, so the tests do not do it
    results = [half(0), half(1)]
    films = []
    for cam, light, stats in results:
        assert stats.completed_iterations == spp // 2 and stats.overflow_flags == 0 and stats.nonfinite_dropped == 0
        assert np.isfinite(cam).all() and np.isfinite(light).all()
        films.append((cam, light))
    return films


@pytest.mark.parametrize("flavour", ["classic", "full", "cloud", "glass"])
def test_bdpt_full_matches_reference_at_4096_spp(etx, golden_dir, flavour):
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, flavour, 4096, {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False})
    golden = load(golden_dir, "cornell_%s_128_bdpt3_4096_rekeyed.npz" % flavour)
    assert int(golden["spp"]) in (4095, 4096)  # CPUBidirectional::update does not count its last iteration (bidirectional.cxx:1526-1531)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], flavour + " bdpt camera+light (independent streams)")
    compare((light_a, light_b), golden["light"], flavour + " bdpt light (independent streams)", mean_limit=1.0e-2, bias_p99_limit=0.2)
    compare((cam_a, cam_b), golden["camera"], flavour + " bdpt camera (independent streams)")
    golden = load(golden_dir, "cornell_%s_128_bdpt3_4096.npz" % flavour)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], flavour + " bdpt camera+light (reference as is)", rmse_limit=2.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.08)


@pytest.mark.parametrize("flavour", ["classic", "full"])
def test_bdpt_shared_streams_match_the_pinned_reference(etx, golden_dir, flavour):
    """As test_gpu_parity_hi.test_vcm_shared_streams_match_the_pinned_reference, for CPUBidirectional (BDPTFull): the device takes the reference's
    seeding - camera path i keeps the seed of light path i (bidirectional.cxx:377-380; option hip-reference_seeding = etx_abi_bdpt_options::reference_seeding, k_bdpt_camera_generate) - and is compared
    with the UNMODIFIED integrator whose film ETX_ORACLE_BVH_DRAWS=opaque_none pins (the same film under every traversal order,
    tests/test_reference_order_spread.py). north_star's limits, no allowance."""
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, flavour, 4096, {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False, "hip-reference_seeding": True})
    golden = load(golden_dir, "cornell_%s_128_bdpt3_4096_opaque_none.npz" % flavour)
    assert int(golden["spp"]) in (4095, 4096)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], flavour + " bdpt camera+light (shared streams, pinned reference)")
    compare((light_a, light_b), golden["light"], flavour + " bdpt light (shared streams, pinned reference)", mean_limit=1.0e-2, bias_p99_limit=0.2)
    compare((cam_a, cam_b), golden["camera"], flavour + " bdpt camera (shared streams, pinned reference)")


@pytest.mark.parametrize("flavour", ["classic", "full", "cloud", "glass"])
def test_bdpt_fast_matches_reference_at_1024_spp(etx, golden_dir, flavour):
    """bdpt-mode BDPTFast, the reference's default (bidirectional.cxx:332): no vertex connections, product-form weights."""
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, flavour, 1024, {"bdpt-mode": etx.api.BDPT_MODE_FAST, "bdpt-blue_noise": False})
    golden = load(golden_dir, "cornell_%s_128_bdpt2_1024_rekeyed.npz" % flavour)
    assert int(golden["spp"]) in (1023, 1024)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], flavour + " bdpt-fast camera+light (independent streams)")
    compare((light_a, light_b), golden["light"], flavour + " bdpt-fast light (independent streams)", mean_limit=1.0e-2, bias_p99_limit=0.2)
    compare((cam_a, cam_b), golden["camera"], flavour + " bdpt-fast camera (independent streams)")
    golden = load(golden_dir, "cornell_%s_128_bdpt2_1024.npz" % flavour)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], flavour + " bdpt-fast camera+light (reference as is)", rmse_limit=2.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.08)


@pytest.mark.parametrize("mode", [0, 1])
def test_bdpt_single_technique_modes(etx, golden_dir, mode):
    """bdpt-mode PathTracing (camera paths + next event estimation only) and LightTracing (light paths splatted to the camera only)."""
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, "full", 1024, {"bdpt-mode": mode, "bdpt-blue_noise": False})
    golden = load(golden_dir, "cornell_full_128_bdpt%d_1024%s.npz" % (mode, "_rekeyed" if mode else ""))
    assert int(golden["spp"]) in (1023, 1024)
    if mode == 0:
        assert float(np.abs(light_a[..., :3]).max()) == 0.0
        compare((cam_a, cam_b), golden["camera"], "full bdpt mode PathTracing camera", rmse_limit=2.0e-3)  # 1024-spp films of a scene with an environment and a sun: the noise estimate itself is that uncertain
    else:
        assert float(np.abs(cam_a[..., :3]).max()) == 0.0
        compare((light_a, light_b), golden["light"], "full bdpt mode LightTracing light", mean_limit=1.0e-2, bias_p99_limit=0.2)


@pytest.mark.parametrize("mode", [2, 3])
def test_bdpt_subsurface_walk_matches_reference(etx, golden_dir, mode):
    """configs[3] family: subsurface materials under the bidirectional integrator. The reference threads the walk through the path
    (bidirectional.cxx:610-633 entry vertex with the scatter material, :746-818 one medium vertex per scattering event, :858-861 exit
    vertex); the device takes such a path out of the wavefront: walk queue, k_bdpt_walk_* for the scattering events, k_bdpt_walk_exit_* for the
    exit vertex (kernels_bdpt.hip). BDPTFull and BDPTFast, 256 spp.
    The entry vertex's connections are attenuated with the medium the reference derives from the SCATTER material (:630-632), which is what
    the vertex-connection comparison of BDPTFull pins."""
    if mode == 3:  # the same mode with the vertex connections off isolates them in a failure
        (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, "sss", 256, {"bdpt-mode": 3, "bdpt-blue_noise": False, "bdpt-conn_connect_vertices": False})
        golden = load(golden_dir, "cornell_sss_128_bdpt3_256_novc_rekeyed.npz")
        compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], "sss bdpt full without vertex connections (independent streams)", rmse_limit=1.5e-3)
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, "sss", 256, {"bdpt-mode": mode, "bdpt-blue_noise": False})
    golden = load(golden_dir, "cornell_sss_128_bdpt%d_256_rekeyed.npz" % mode)
    assert int(golden["spp"]) in (255, 256)
    label = "sss bdpt mode %d " % mode
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], label + "camera+light (independent streams)", rmse_limit=1.5e-3)
    compare((light_a, light_b), golden["light"], label + "light (independent streams)", rmse_limit=1.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.2)
    compare((cam_a, cam_b), golden["camera"], label + "camera (independent streams)", rmse_limit=1.5e-3)
    golden = load(golden_dir, "cornell_sss_128_bdpt%d_256.npz" % mode)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], label + "camera+light (reference as is)", rmse_limit=2.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.08)


def test_bdpt_spectral_scene_matches_reference(etx, golden_dir, cie_observer):
    """configs[2]'s scene family under the bidirectional integrator: one wavelength per path (the camera path reuses the light path's,
    bidirectional.cxx:377-391), dispersive dielectrics, a rough conductor; 2 892 triangles on the tree. BDPTFull, 1024 spp."""
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, "gems", 1024, {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False}, cie=cie_observer)
    golden = load(golden_dir, "cornell_gems_128_bdpt3_1024_rekeyed.npz")
    assert int(golden["spp"]) in (1023, 1024)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], "gems bdpt camera+light (independent streams)", rmse_limit=1.5e-3)
    compare((cam_a, cam_b), golden["camera"], "gems bdpt camera (independent streams)", rmse_limit=1.5e-3)
    golden = load(golden_dir, "cornell_gems_128_bdpt3_1024.npz")
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], "gems bdpt camera+light (reference as is)", rmse_limit=2.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.08)


def test_bdpt_spectral_subsurface_walk_matches_reference(etx, golden_dir, cie_observer):
    """The subsurface box in spectral mode: the medium a walk runs through is derived from the material's colour and scattering distances
    AT THE PATH'S WAVELENGTH (subsurface_step, bidirectional.cxx:757-771; device: DMedium::derived_color / _distances + medium_coefficients)."""
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, "sssspec", 256, {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False}, cie=cie_observer)
    golden = load(golden_dir, "cornell_sssspec_128_bdpt3_256_rekeyed.npz")
    assert int(golden["spp"]) in (255, 256)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], "sss spectral bdpt camera+light (independent streams)", rmse_limit=1.5e-3)
    compare((cam_a, cam_b), golden["camera"], "sss spectral bdpt camera (independent streams)", rmse_limit=1.5e-3)
    golden = load(golden_dir, "cornell_sssspec_128_bdpt3_256.npz")
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], "sss spectral bdpt camera+light (reference as is)", rmse_limit=2.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.08)


def test_bdpt_rejects_what_it_does_not_implement(etx, golden_dir, cie_observer):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_classic_128.etxscene"))
    integ = etx.HIPBidirectional(snap)
    integ.options().update({"bdpt-mode": 7})
    with pytest.raises(etx.EtxHipError, match="bdpt-mode"):
        integ.run()
    integ.context.close()
