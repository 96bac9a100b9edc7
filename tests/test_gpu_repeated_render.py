"""GPU (-m gpu): repeated renders of the general-material scenes in ONE process.

Round 1 hid a build-dependent NaN in the general-material VCM kernels ("the first render of a process is clean, later ones
are not", "the count depends on the lane count"): kernels at the register allocator's limit, caught only by accident.
The general kernels now call the BSDF classes out of line (dev_bsdf_ool.h) and every film write is guarded
(dev_vcm.h film_value_ok). This test is the guard for exactly that failure mode: rough / glass / gems, three renders per
process, one and four lanes, 256 samples per pixel - raw film sums finite, no contribution dropped as non-finite, and the
blue channel (where the NaN pixels sat) within 1 % of the reference's film.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPP = 256


def reference_film(golden_dir, flavour):
    """The reference's CPUVCM film of the FIRST iterations (64 or 256 spp, oracle/gen_golden.py). Not the 4096-spp film: the merge
    radius shrinks with the iteration index (vcm_cpu.cxx:100-113), and with it the merging bias - in the gems box the blue mean is
    0.0092 over iterations 0-63, 0.0091 over 0-255, 0.0076 over 2048-2303 (device) against 0.0092 / 0.0078 (reference, 64 / 4096 spp)."""
    golden = np.load(os.path.join(golden_dir, "cornell_%s_128_vcm.npz" % flavour))
    total = golden["camera"] + golden["light"]
    ok = np.isfinite(total).all(axis=2)  # the reference's release build lets an occasional NaN sample through
    assert ok.mean() > 0.999
    return total, ok, int(golden["spp"])


@pytest.mark.parametrize("lanes", [1, 4])
@pytest.mark.parametrize("flavour", ["rough", "glass", "gems"])
def test_general_materials_stay_finite_over_repeated_renders(etx, golden_dir, cie_observer, monkeypatch, flavour, lanes):
    monkeypatch.setenv("ETX_HIP_LANES", str(lanes))  # read by etx_hip_create
    reference, ok, reference_spp = reference_film(golden_dir, flavour)
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_%s_128.etxscene" % flavour))
    snap.samples = SPP
    integ = etx.HIPVCM(snap)
    integ.options()["vcm-blue_noise"] = False
    if flavour == "gems":
        integ.cie_table = cie_observer
    means = []
    for repeat in range(3):
        integ.render()
        cam = integ.film(etx.api.LAYER_CAMERA)
        light = integ.film(etx.api.LAYER_LIGHT)
        stats = integ.status()
        assert stats.completed_iterations == SPP and stats.overflow_flags == 0
        assert stats.nonfinite_dropped == 0, "render %d dropped %d non-finite film contributions" % (repeat, stats.nonfinite_dropped)
        assert np.isfinite(cam).all() and np.isfinite(light).all(), "render %d: non-finite raw film sums" % repeat
        total = (cam + light)[..., :3]
        means.append(np.where(ok[..., None], total, 0.0).mean(axis=(0, 1)))
    integ.context.close()
    ref_mean = np.where(ok[..., None], reference, 0.0).mean(axis=(0, 1))
    for mean in means:
        rel = (mean - ref_mean) / ref_mean
        # blue: the channel the round-1 NaN pixels were in (-2.6 ... -4.8 % then). Against a 64-spp film of the gems scene,
        # whose blue mean is 0.004, the reference's own noise allows 2 %
        assert abs(rel[2]) < (1.0e-2 if (reference_spp >= 1024 or flavour != "gems") else 2.0e-2), (flavour, lanes, rel)
        assert np.abs(rel).max() < 2.0e-2, (flavour, lanes, rel)
    # the three renders are the same iterations of the same scene: identical up to the order of the float atomics
    assert np.abs(means[1] - means[0]).max() < 2.0e-3 * ref_mean.max()
    assert np.abs(means[2] - means[0]).max() < 2.0e-3 * ref_mean.max()
