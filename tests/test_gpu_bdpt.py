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

    # one after the other (contexts driven from several host threads at once are tests/test_gpu_contexts.py's subject)
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


def test_bdpt_textured_subsurface_walk_matches_reference(etx, golden_dir):
    """A subsurface material WITHOUT an interior medium whose scattering colour is TEXTURED (checker through map_Kd on the short box of the subsurface
    Cornell box, scenes/make_scenes.py `ssstex`): the reference derives the medium of a walk at the ENTRY POINT (subsurface_step,
    bidirectional.cxx:757-771: apply_image at intersection.tex, subsurface::remap per channel). Until round 5 etx_hip_begin refused such a scene for this
    integrator; now every walk appends a row of its own to the lane's copy of the medium table (bdpt_derive_walk_medium, kernels_bdpt.hip). BDPTFull, 256 spp,
    the limits of the untextured box (test_bdpt_subsurface_walk_matches_reference)."""
    (cam_a, light_a), (cam_b, light_b) = render_halves(etx, golden_dir, "ssstex", 256, {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False})
    golden = load(golden_dir, "cornell_ssstex_128_bdpt3_256_rekeyed.npz")
    assert int(golden["spp"]) in (255, 256)
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], "textured sss bdpt camera+light (independent streams)", rmse_limit=1.5e-3)
    compare((light_a, light_b), golden["light"], "textured sss bdpt light (independent streams)", rmse_limit=1.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.2)
    compare((cam_a, cam_b), golden["camera"], "textured sss bdpt camera (independent streams)", rmse_limit=1.5e-3)
    golden = load(golden_dir, "cornell_ssstex_128_bdpt3_256.npz")
    compare((cam_a + light_a, cam_b + light_b), golden["camera"] + golden["light"], "textured sss bdpt camera+light (reference as is)", rmse_limit=2.5e-3, mean_limit=1.0e-2, bias_p99_limit=0.08)
    # the texture is in the film: the two colours of the checker give the short box's top face a contrast the untextured box does not have
    untextured = load(golden_dir, "cornell_sss_128_bdpt3_256_rekeyed.npz")
    def blocks(film):
        return (film["camera"] + film["light"])[..., :3].reshape(16, 8, 16, 8, 3).mean(axis=(1, 3))
    assert float(np.abs(blocks(golden) - blocks(untextured)).max()) > 1.0e-2, "the textured golden film equals the untextured one (two renders of one scene differ by 1.6e-3 at most): the fixture does not exercise the per-walk media"


@pytest.mark.parametrize("flavour", ["sss", "ssstex"])
def test_bdpt_mixed_scene_split_by_bsdf_class_renders_the_same_film(etx, golden_dir, flavour):
    """A scene that mixes Lambert surfaces with a material of a general BSDF class (the plastic coat of the tall subsurface box): since round 6 the
    bidirectional kernels run their inline instantiation over every item and the out-of-line one over the items of general classes only (kernels_bdpt.hip
    kPartSimple / kPartGeneral); debug flag 0x10000 restores the general instantiation for every item. Every path draws from its own stream and the class a
    surface is shaded by does not depend on the kernel that runs it, so the two films are the same up to the order of the film's float additions."""
    def render(flags):
        snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_%s_128.etxscene" % flavour))
        snap.samples = 16
        integ = etx.HIPBidirectional(snap)
        integ.options().update({"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False})
        if flags:
            integ.context.set_debug_flags(flags)
        integ.render()
        cam, light = integ.film(etx.api.LAYER_CAMERA)[..., :3].astype(np.float64), integ.film(etx.api.LAYER_LIGHT)[..., :3].astype(np.float64)
        stats = integ.status()
        integ.context.close()
        assert stats.completed_iterations == 16 and stats.overflow_flags == 0 and stats.nonfinite_dropped == 0
        return cam, light
    cam_split, light_split = render(0)
    cam_all, light_all = render(0x10000)
    assert float(cam_split.mean()) > 1.0e-2 and float(light_split.mean()) > 1.0e-3
    # (the inline and the out-of-line Lambert code may round a sampled direction differently in the last bit: a handful of paths then take another discrete
    # decision somewhere - a pixel here and there differs by one path's contribution. A surface shaded by the wrong class would change a whole object.)
    for name, a, b in (("camera", cam_split, cam_all), ("light", light_split, light_all)):
        scale = float(np.abs(b).mean())
        different = float((np.abs(a - b).max(axis=-1) > 1.0e-3 * scale).mean())
        assert different <= 5.0e-3, "%s %s layer: %.2f %% of the pixels of the split film differ from the unsplit one" % (flavour, name, 100.0 * different)
        assert abs(float(a.mean()) - float(b.mean())) <= 2.0e-3 * scale, (flavour, name, float(a.mean()), float(b.mean()))
