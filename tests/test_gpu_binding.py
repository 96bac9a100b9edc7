"""GPU (-m gpu): the reference-side C++ binding (integration/etx_hip_integrators.hxx: HIPVCM / HIPPathTracing on the
reference's `struct Integrator`, compiled against the reference's headers into oracle/_ref/etx_oracle) against the ctypes
binding of the same C ABI. Both drive libetx_hip.so; the C++ side publishes the film through the reference's own Film
(accumulate_camera_image / atomic_add_light_iteration / commit_light_iteration) and the driver writes Film::layer."""
import os
import subprocess

import numpy as np
import pytest

from tools import film_io

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")
HIP_RENDER = os.path.join(ROOT, "oracle", "_ref", "etx_hip_render")  # the same driver without the reference's CPU integrators: the backend's own headless host


def run_driver(tmp_path, snapshot, integrator, spp, *options, extra=(), name=None, binary=ORACLE):
    out = str(tmp_path / ("%s.raw" % (name or integrator)))
    cmd = [binary, "--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", out] + list(extra)
    for o in options:
        cmd += ["--opt", o]
    result = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert result.returncode == 0, result.stdout[-2000:]
    return film_io.read_film(out), result.stdout


def test_cpp_hipvcm_matches_ctypes_binding(etx, golden_dir, tmp_path, bluenoise_64spp):
    snapshot = os.path.join(golden_dir, "cornell_full_128.etxscene")
    spp = 64  # scene.samples of the snapshot: VCMOptions defaults (blue noise on), the binding tabulates BNSampler itself
    film, log = run_driver(tmp_path, snapshot, "hip-vcm", spp)
    assert film["spp"] == spp and "VCM (HIP gfx950)" in log
    snap = etx.SceneSnapshot(snapshot)
    integ = etx.HIPVCM(snap)
    integ.bluenoise_tables = {6: bluenoise_64spp}
    integ.render()
    cam, light, res = (integ.film(getattr(etx.api, "LAYER_" + n)) for n in ("CAMERA", "LIGHT", "RESULT"))
    integ.context.close()
    # same iterations, same kernels: the films differ only by the order of the float atomics
    np.testing.assert_allclose(film["camera"][..., :3], cam[..., :3], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(film["light"][..., :3], light[..., :3], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(film["result"][..., :3], res[..., :3], rtol=2e-4, atol=4e-5)  # Film::layer(Result) of the reference


def test_cpp_binding_reference_seeding_option(etx, golden_dir, tmp_path):
    """`hip-reference_seeding` is an option key of the compiled binding (integration/etx_hip_integrators.hxx), next to the reference integrator's
    own keys: the host asks for the reference's seeding of the camera path (vcm_shared.hxx:312,357) through the same Options store, and the
    byte travels in etx_abi_vcm_options::reference_seeding. Same films as the ctypes path with the same option, another film than the default."""
    snapshot = os.path.join(golden_dir, "cornell_full_128.etxscene")
    seeded, _ = run_driver(tmp_path, snapshot, "hip-vcm", 16, "vcm-blue_noise=false", "hip-reference_seeding=true", name="seeded")
    default, _ = run_driver(tmp_path, snapshot, "hip-vcm", 16, "vcm-blue_noise=false", name="default")
    snap = etx.SceneSnapshot(snapshot)
    snap.samples = 16
    integ = etx.HIPVCM(snap)
    integ.options().update({"vcm-blue_noise": False, "hip-reference_seeding": True})
    integ.render()
    cam, light = integ.film(etx.api.LAYER_CAMERA), integ.film(etx.api.LAYER_LIGHT)
    integ.context.close()
    np.testing.assert_allclose(seeded["camera"][..., :3], cam[..., :3], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(seeded["light"][..., :3], light[..., :3], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(seeded["light"][..., :3], default["light"][..., :3], rtol=2e-4, atol=2e-5)  # the light paths are seeded alike in both
    assert np.abs(seeded["camera"][..., :3] - default["camera"][..., :3]).max() > 1.0e-2                  # the camera paths are not


def test_cpp_hip_path_tracer_matches_ctypes_binding(etx, golden_dir, tmp_path):
    snapshot = os.path.join(golden_dir, "cornell_rough_128.etxscene")
    film, log = run_driver(tmp_path, snapshot, "hip-pt", 16, "bn=false")
    assert "Path Tracing (HIP gfx950)" in log
    snap = etx.SceneSnapshot(snapshot)
    snap.samples = 16
    integ = etx.HIPPathTracing(snap)
    integ.options()["bn"] = False
    integ.render()
    layers = {n: integ.film(getattr(etx.api, "LAYER_" + n.upper())) for n in ("camera", "normal", "albedo")}
    integ.context.close()
    np.testing.assert_allclose(film["camera"][..., :3], layers["camera"][..., :3], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(film["normal"][..., :3], layers["normal"][..., :3], rtol=0, atol=1e-5)  # Film::layer(Normals) = n * 0.5 + 0.5
    np.testing.assert_allclose(film["albedo"][..., :3], layers["albedo"][..., :3], rtol=2e-4, atol=2e-5)
    assert np.abs(film["light"][..., :3]).max() == 0.0


def test_cpp_driver_checkpoint_and_resume(golden_dir, tmp_path):
    """The headless driver of the HIP path: --max-iterations + --checkpoint stops a render and stores its film state,
    --resume continues it in a new process; the result is the uninterrupted render (same iterations, float addition order aside)."""
    snapshot = os.path.join(golden_dir, "cornell_full_128.etxscene")
    assert subprocess.run([HIP_RENDER, "--load-snapshot", snapshot, "--integrator", "vcm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode == 1  # no CPU integrator in this binary
    whole, _ = run_driver(tmp_path, snapshot, "hip-vcm", 32, "vcm-blue_noise=false", name="whole", binary=HIP_RENDER)
    checkpoint = str(tmp_path / "render.etxc")
    part, _ = run_driver(tmp_path, snapshot, "hip-vcm", 32, "vcm-blue_noise=false", extra=["--max-iterations", "8", "--checkpoint", checkpoint], name="part", binary=HIP_RENDER)
    assert 8 <= part["spp"] < 24 and os.path.getsize(checkpoint) > 128 * 128 * 64  # the lanes' iterations in flight finish, the rest is left
    resumed, log = run_driver(tmp_path, snapshot, "hip-vcm", 32, "vcm-blue_noise=false", extra=["--resume", checkpoint], name="resumed", binary=HIP_RENDER)
    assert resumed["spp"] == 32 and ("resumed at iteration %d" % part["spp"]) in log
    np.testing.assert_allclose(resumed["camera"][..., :3], whole["camera"][..., :3], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(resumed["light"][..., :3], whole["light"][..., :3], rtol=2e-4, atol=2e-5)
    assert np.abs(part["camera"][..., :3] - whole["camera"][..., :3]).max() > 1.0e-3


def test_cpp_hip_bidirectional_matches_ctypes_binding(etx, golden_dir, tmp_path):
    snapshot = os.path.join(golden_dir, "cornell_full_128.etxscene")
    film, log = run_driver(tmp_path, snapshot, "hip-bdpt", 16, "bdpt-blue_noise=false", "bdpt-mode=3")
    assert "Bidirectional (HIP gfx950)" in log
    snap = etx.SceneSnapshot(snapshot)
    snap.samples = 16
    integ = etx.HIPBidirectional(snap)
    integ.options().update({"bdpt-blue_noise": False, "bdpt-mode": etx.api.BDPT_MODE_FULL})
    integ.render()
    cam, light = integ.film(etx.api.LAYER_CAMERA), integ.film(etx.api.LAYER_LIGHT)
    integ.context.close()
    np.testing.assert_allclose(film["camera"][..., :3], cam[..., :3], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(film["light"][..., :3], light[..., :3], rtol=2e-4, atol=2e-5)


def test_cpp_binding_spectral_scene_uploads_the_observer(etx, golden_dir, tmp_path, cie_observer):
    snapshot = os.path.join(golden_dir, "cornell_diamond_128.etxscene")
    film, _ = run_driver(tmp_path, snapshot, "hip-vcm", 8, "vcm-blue_noise=false")
    snap = etx.SceneSnapshot(snapshot)
    snap.samples = 8
    integ = etx.HIPVCM(snap)
    integ.options()["vcm-blue_noise"] = False
    integ.cie_table = cie_observer
    integ.render()
    cam = integ.film(etx.api.LAYER_CAMERA)
    integ.context.close()
    np.testing.assert_allclose(film["camera"][..., :3], cam[..., :3], rtol=5e-4, atol=5e-5)


def test_cpp_binding_publishes_the_preview_at_pixel_size_8(golden_dir, tmp_path):
    """Film::pixel_size() > 1 - the GUI sets 8 while the camera moves (app.cxx:135) and the CPU integrators then render one path per
    block (film.cxx:152-168, 185-199). The device renders the full frame regardless; the binding publishes block means through the
    same two Film calls, which fill the blocks: the film shows the current render, coarse, as the reference's preview does."""
    snapshot = os.path.join(golden_dir, "cornell_full_128.etxscene")
    fine, _ = run_driver(tmp_path, snapshot, "hip-vcm", 16, "vcm-blue_noise=false", name="fine")
    coarse, _ = run_driver(tmp_path, snapshot, "hip-vcm", 16, "vcm-blue_noise=false", extra=["--pixel-size", "8"], name="coarse")
    assert coarse["spp"] == 16
    for layer in ("camera", "light"):
        blocks = coarse[layer][..., :3].reshape(16, 8, 16, 8, 3)
        assert np.abs(blocks - blocks[:, :1, :, :1]).max() == 0.0, layer  # every 8 x 8 block holds one value
        expected = fine[layer][..., :3].reshape(16, 8, 16, 8, 3).mean(axis=(1, 3))
        np.testing.assert_allclose(blocks[:, 0, :, 0], expected, rtol=1e-3, atol=1e-4, err_msg=layer)  # ... the mean of the full render
    assert float(coarse["camera"][..., :3].mean()) > 0.01
