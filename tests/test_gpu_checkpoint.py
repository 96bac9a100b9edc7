"""GPU (-m gpu): checkpoint / resume of a render in progress (etx_hip_checkpoint_save / _load).

The reference has no checkpoint - a stopped render starts over (app.cxx:193-216) - so the property tested is the one that makes a
checkpoint worth having: an interrupted render that is resumed in a NEW context is the uninterrupted render. Iterations are seeded by
(pixel, iteration) and carry their own radius / MIS weights (vcm_cpu.cxx:100-113), so the two films hold the same per-iteration terms and
differ only by the order of the float additions (lanes commit in completion order, light splats are atomics).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(etx, golden_dir, cls, flavour, spp, options, noise_threshold=None):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_%s_128.etxscene" % flavour))
    snap.samples = spp
    if noise_threshold is not None:
        snap.noise_threshold = noise_threshold
    integ = cls(snap)
    integ.options().update(options)
    return integ


def films(etx, integ):
    return integ.film(etx.api.LAYER_CAMERA)[..., :3].astype(np.float64), integ.film(etx.api.LAYER_LIGHT)[..., :3].astype(np.float64)


def close_films(a, b, label):
    # the same terms added in a different order: float32 sums of <= 64 terms of mixed magnitude
    scale = max(float(a.mean()), 1.0e-6)
    worst = float(np.abs(a - b).max())
    assert worst <= 2.0e-4 * max(float(a.max()), scale), "%s: max abs difference %g (mean %g, max %g)" % (label, worst, scale, float(a.max()))
    assert abs(float(a.mean()) - float(b.mean())) <= 1.0e-5 * scale, "%s: means %g vs %g" % (label, a.mean(), b.mean())


@pytest.mark.parametrize("kind", ["vcm", "bdpt", "pt"])
def test_resumed_render_is_the_uninterrupted_render(etx, golden_dir, tmp_path, kind):
    cls = {"vcm": etx.HIPVCM, "bdpt": etx.HIPBidirectional, "pt": etx.HIPPathTracing}[kind]
    options = {"vcm": {"vcm-blue_noise": False}, "bdpt": {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False}, "pt": {"bn": False}}[kind]
    spp, cut = 64, 24
    whole = make(etx, golden_dir, cls, "full", spp, options, noise_threshold=0.0)
    whole.render()
    cam_w, light_w = films(etx, whole)
    assert whole.status().completed_iterations == spp
    whole.context.close()

    first = make(etx, golden_dir, cls, "full", spp, options, noise_threshold=0.0)
    first.run()
    while first._rendered < cut:
        first.update()
    path = str(tmp_path / "render.etxc")
    blob = first.save_checkpoint(path)
    saved_at = first.status().completed_iterations
    assert saved_at == cut and os.path.getsize(path) == len(blob)
    cam_cut, _ = films(etx, first)
    first.context.close()

    second = make(etx, golden_dir, cls, "full", spp, options, noise_threshold=0.0)
    second.resume(path)
    assert second.status().completed_iterations == cut and second.state() == etx.integrator.State.Running
    cam_resumed, _ = films(etx, second)
    assert np.array_equal(cam_resumed, cam_cut)  # the film of the checkpoint, bit for bit, before anything else is rendered
    second.finish()
    stats = second.status()
    assert stats.completed_iterations == spp and stats.overflow_flags == 0 and stats.nonfinite_dropped == 0
    cam_r, light_r = films(etx, second)
    second.context.close()
    close_films(cam_w, cam_r, kind + " camera image")
    if kind != "pt":
        close_films(light_w, light_r, kind + " light image")
    assert float(np.abs(cam_w - cam_cut).max()) > 1.0e-3  # the comparison above is not trivially true: 24 and 64 spp differ


def test_checkpoint_crosses_processes(etx, golden_dir, tmp_path):
    """A checkpoint written by ANOTHER process resumes here: the header's scene hash is taken field by field (host_scene.cpp content_hash), so nothing
    of it depends on what a process left in the padding of the scene's structs."""
    import subprocess
    import sys
    spp, cut = 32, 12
    path = str(tmp_path / "child.etxc")
    snapshot = os.path.join(golden_dir, "cornell_classic_128.etxscene")
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "checkpoint_child.py")
    subprocess.run([sys.executable, child, snapshot, str(spp), str(cut), path], check=True, timeout=300)
    whole = make(etx, golden_dir, etx.HIPVCM, "classic", spp, {"vcm-blue_noise": False})
    whole.render()
    cam_w, light_w = films(etx, whole)
    whole.context.close()
    second = make(etx, golden_dir, etx.HIPVCM, "classic", spp, {"vcm-blue_noise": False})
    second.resume(path)
    assert second.status().completed_iterations == cut
    second.finish()
    assert second.status().completed_iterations == spp
    cam_r, light_r = films(etx, second)
    second.context.close()
    close_films(cam_w, cam_r, "cross-process camera image")
    close_films(light_w, light_r, "cross-process light image")


def test_checkpoint_keeps_the_adaptive_sampling_state(etx, golden_dir):
    """Path tracing with Scene::noise_threshold: the even-sample sums, the per-pixel sample counts and the converged flags travel with
    the film, so the resumed render goes on sampling exactly the pixels the interrupted one would have."""
    spp, cut = 96, 48
    whole = make(etx, golden_dir, etx.HIPPathTracing, "classic", spp, {"bn": False}, noise_threshold=0.05)
    whole.render()
    cam_w, _ = films(etx, whole)
    samples_w = whole.status().active_pixels
    whole.context.close()

    first = make(etx, golden_dir, etx.HIPPathTracing, "classic", spp, {"bn": False}, noise_threshold=0.05)
    first.run()
    while first._rendered < cut:
        first.update()
    blob = first.save_checkpoint()
    samples_cut = first.status().active_pixels
    first.context.close()

    second = make(etx, golden_dir, etx.HIPPathTracing, "classic", spp, {"bn": False}, noise_threshold=0.05)
    second.resume(blob).finish()
    cam_r, _ = films(etx, second)
    samples_r = samples_cut + second.status().active_pixels
    second.context.close()
    assert samples_w < spp * 128 * 128  # the threshold did stop pixels, or the test tests nothing
    # an adaptive render runs on one device lane, every iteration reads the mask its predecessor left: the resumed render samples
    # exactly the pixels the uninterrupted one did
    assert int(samples_r) == int(samples_w), (samples_r, samples_w)
    assert abs(float(cam_r.mean()) - float(cam_w.mean())) <= 1.0e-5 * float(cam_w.mean())


def test_checkpoint_of_another_run_is_refused(etx, golden_dir):
    first = make(etx, golden_dir, etx.HIPVCM, "classic", 8, {"vcm-blue_noise": False})
    with pytest.raises(etx.EtxHipError):
        first.save_checkpoint()  # nothing armed yet
    first.render()
    blob = first.save_checkpoint()
    first.context.close()

    other_options = make(etx, golden_dir, etx.HIPVCM, "classic", 8, {"vcm-blue_noise": False, "vcm-merge_vertices": False})
    with pytest.raises(etx.EtxHipError, match="different run"):
        other_options.resume(blob)
    other_options.context.close()

    other_scene = make(etx, golden_dir, etx.HIPVCM, "glass", 8, {"vcm-blue_noise": False})  # same film size and options, other materials
    other_length = make(etx, golden_dir, etx.HIPVCM, "classic", 8, {"vcm-blue_noise": False})
    other_length.snapshot.max_path_length = 5
    for other in (other_scene, other_length):
        with pytest.raises(etx.EtxHipError, match="different run"):
            other.resume(blob)
        other.context.close()

    other_integrator = make(etx, golden_dir, etx.HIPPathTracing, "classic", 8, {"bn": False})
    with pytest.raises(etx.EtxHipError, match="different run"):
        other_integrator.resume(blob)
    with pytest.raises(etx.EtxHipError, match="not a checkpoint"):
        other_integrator.resume(b"\0" * 64)
    other_integrator.context.close()

    sharded = etx.HIPVCM(etx.SceneSnapshot(os.path.join(golden_dir, "cornell_classic_128.etxscene")), first_iteration=1, iteration_stride=2)
    sharded.options()["vcm-blue_noise"] = False
    with pytest.raises(etx.EtxHipError, match="different run"):
        sharded.resume(blob)
    sharded.context.close()


def test_working_set_follows_the_integrators_in_use(etx, golden_dir):
    """etx_hip_device_bytes: the photon grid belongs to VCM and the fifth / sixth lane to the bidirectional integrator - a context that
    never runs them never allocates them, one that switches integrators (the reference's GUI does, app.cxx integrator list) renders the
    same films afterwards."""
    if os.environ.get("ETX_HIP_LANES"):
        pytest.skip("lane counts fixed by ETX_HIP_LANES")
    pt = make(etx, golden_dir, etx.HIPPathTracing, "full", 8, {"bn": False}, noise_threshold=0.0)
    pt.render()
    pt_bytes = pt.context.device_bytes()
    pt.context.close()

    bdpt = make(etx, golden_dir, etx.HIPBidirectional, "full", 16, {"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False})
    bdpt.render()
    bdpt_bytes = bdpt.context.device_bytes()
    cam_b, light_b = films(etx, bdpt)
    assert bdpt_bytes > pt_bytes * 1.3, "six lanes against four: %d vs %d bytes" % (bdpt_bytes, pt_bytes)

    # the same context, now VCM: the grid arrives for the four lanes VCM uses; then the bidirectional render again
    vcm = etx.HIPVCM(bdpt.snapshot)
    vcm.context.close()
    vcm.context, vcm._uploaded_version = bdpt.context, bdpt._uploaded_version  # the scene is on the device already
    vcm.options().update({"vcm-blue_noise": False})
    vcm.render()
    vcm_bytes = bdpt.context.device_bytes()
    assert vcm_bytes > bdpt_bytes, "photon grid: %d vs %d bytes" % (vcm_bytes, bdpt_bytes)
    bdpt.render()
    assert bdpt.context.device_bytes() == vcm_bytes
    cam_b2, light_b2 = films(etx, bdpt)
    close_films(cam_b, cam_b2, "bdpt camera film after a VCM render in the same context")
    close_films(light_b, light_b2, "bdpt light film after a VCM render in the same context")
    bdpt.context.close()
