"""GPU (-m gpu): scene edits without a full upload (etx_hip_update_scene).

The reference re-commits the whole scene on every change (Raytracing::commit_changes rebuilds the Embree BVH, rt.cxx:58-88, app.cxx:368-399).
The device path keeps geometry, BVH and images resident: moved vertices refit the BVH4 on the device (kernels_bvh_build.hip), edited
materials rebuild the small tables and the traversal filters. The property tested: an updated context is indistinguishable from a
context that uploaded the edited scene from scratch - same closest hits (brute-force oracle beside it), same images.
"""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import make_rays
from tools import synthetic_scenes

pytestmark = pytest.mark.gpu

ETX_MAT_MIRROR, ETX_MAT_VOID = 6, 10


def move_material(snap, material, offset):
    """translates every vertex of the triangles with that material (the gems scene does not share vertices between objects)"""
    triangles, vertices = snap.triangles(), snap.vertices()
    corners = np.unique(triangles[triangles[:, 3] == material][:, 0:3].reshape(-1))
    vertices[corners, 0:3] += np.asarray(offset, dtype=np.float32)
    return len(corners)


def hit_triangles(hits):
    tri = hits[:, 3].view(np.uint32).astype(np.int64)
    tri[tri == 0xFFFFFFFF] = -1
    return tri


def gem_material(snap):
    """the dielectric material with the most triangles"""
    triangles, classes = snap.triangles(), snap.material_classes()
    counts = {m: int((triangles[:, 3] == m).sum()) for m in range(len(classes)) if classes[m] == 4}
    return max(counts, key=counts.get)


def test_refit_after_moved_vertices_equals_fresh_upload(etx, golden_dir):
    from oracle import ray_oracle
    path = os.path.join(golden_dir, "cornell_gems_128.etxscene")
    snap = etx.SceneSnapshot(path)
    rays = make_rays(60000, 11)
    updated = etx.api.Context(0)
    updated.upload_scene(snap)
    before = updated.trace_rays(rays)
    assert move_material(snap, gem_material(snap), (0.11, 0.07, -0.05)) > 300
    updated.update_scene(snap, etx.api.CHANGED_POSITIONS)
    after = updated.trace_rays(rays)
    fresh = etx.api.Context(0)
    fresh.upload_scene(snap)  # host SAH build over the moved vertices
    reference = fresh.trace_rays(rays)
    # a closest hit does not depend on the tree: the refit tree and the rebuilt tree agree hit for hit (ties between coplanar
    # triangles aside), and with the brute-force intersection of the moved triangles
    same = hit_triangles(after) == hit_triangles(reference)
    assert same.mean() > 0.9995
    np.testing.assert_allclose(after[same, 2], reference[same, 2], rtol=1e-6, atol=1e-6)
    sample = slice(0, 4000)
    brute = ray_oracle.closest_hits(snap, rays[sample])
    assert (hit_triangles(after[sample]) == brute[:, 3].astype(np.int64)).mean() > 0.999
    assert (hit_triangles(after) != hit_triangles(before)).mean() > 0.01  # the edit is visible to the rays
    # moving it back restores the original hits exactly (the refit boxes are the builder's boxes: same vertices, same min / max)
    restored = etx.SceneSnapshot(path)  # (adding and subtracting an offset in float need not round-trip: the file's own vertices)
    updated.update_scene(restored, etx.api.CHANGED_POSITIONS)
    again = updated.trace_rays(rays)
    np.testing.assert_array_equal(again.view(np.uint32), before.view(np.uint32))
    updated.close()
    fresh.close()


def render(etx, integ):
    integ.render()
    stats = integ.status()
    assert stats.overflow_flags == 0 and stats.nonfinite_dropped == 0
    return integ.film(etx.api.LAYER_CAMERA)[..., :3].astype(np.float64), integ.film(etx.api.LAYER_LIGHT)[..., :3].astype(np.float64)


def assert_same_render(a, b, label):
    for x, y, layer in ((a[0], b[0], "camera"), (a[1], b[1], "light")):
        scale = max(float(np.abs(x).mean()), 1.0e-6)
        assert abs(float(x.mean()) - float(y.mean())) <= 2.0e-5 * scale, "%s %s: means %g vs %g" % (label, layer, x.mean(), y.mean())
        assert float(np.abs(x - y).max()) <= 5.0e-4 * max(float(x.max()), scale), "%s %s: max abs difference %g" % (label, layer, np.abs(x - y).max())


@pytest.mark.parametrize("scene,edit,kind", [("gems", "positions", "pt"), ("gems", "material", "pt"), ("classic", "material", "vcm"), ("full", "positions", "vcm")])
def test_integrator_renders_the_edited_scene(etx, golden_dir, cie_observer, scene, edit, kind):
    """scene_edited + run(): the second render of an integrator whose scene was edited in place equals the render of a new
    integrator on the edited scene (same iterations and seeds; float addition order aside). gems: BVH refit / filter update;
    classic, full: scenes of the flat sweep, whose pre-transformed primitives are rebuilt on the host. gems is rendered by the path
    tracer: its rough conductor evaluates stochastically (Heitz walk), and VCM seeds those evaluations by pool position - two VCM
    renders of that scene agree statistically, not value by value (tools/determinism_probe.py)."""
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_%s_128.etxscene" % scene))
    snap.samples = 16

    def integrator():
        integ = (etx.HIPVCM if kind == "vcm" else etx.HIPPathTracing)(snap)
        integ.options().update({"vcm-blue_noise": False} if kind == "vcm" else {"bn": False})
        integ.cie_table = cie_observer  # gems is a spectral scene
        return integ

    integ = integrator()
    original = render(etx, integ)
    if edit == "positions":
        triangles, classes = snap.triangles(), snap.material_classes()
        boxes = [m for m in range(len(classes)) if (classes[m] == 0) and (int((triangles[:, 3] == m).sum()) == 10)]  # full: the diffuse box
        material = gem_material(snap) if scene == "gems" else boxes[0]
        move_material(snap, material, (0.08, 0.0, 0.06) if scene == "gems" else (0.0, 0.05, 0.0))
        integ.scene_edited(etx.api.CHANGED_POSITIONS)
    else:
        material = int(snap.triangles()[0, 3])  # the material of the first triangle (a wall): becomes a mirror
        assert snap.material_classes()[material] == 0
        snap.materials()[material, 41] = ETX_MAT_MIRROR
        integ.scene_edited(etx.api.CHANGED_MATERIALS)
    edited = render(etx, integ)
    integ.context.close()
    fresh_integ = integrator()
    fresh = render(etx, fresh_integ)
    fresh_integ.context.close()
    assert_same_render(edited, fresh, "%s %s" % (scene, edit))
    assert float(np.abs(edited[0] - original[0]).mean()) > 1.0e-3 * float(original[0].mean())  # and it is not the unedited scene


def test_void_material_updates_the_traversal_filter(etx, golden_dir):
    """Material::Class::Void triangles are never reported (rt.cxx:441-444): the class lives in the traversal triangles' flags, which
    etx_hip_update_scene re-derives on the device."""
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_gems_128.etxscene"))
    rays = make_rays(30000, 5)
    ctx = etx.api.Context(0)
    ctx.upload_scene(snap)
    material = gem_material(snap)
    gem_triangles = np.nonzero(snap.triangles()[:, 3] == material)[0]
    before = hit_triangles(ctx.trace_rays(rays))
    assert np.isin(before, gem_triangles).mean() > 0.005
    snap.materials()[material, 41] = ETX_MAT_VOID
    ctx.update_scene(snap, etx.api.CHANGED_MATERIALS)
    after = hit_triangles(ctx.trace_rays(rays))
    assert np.isin(after, gem_triangles).sum() == 0
    untouched = ~np.isin(before, gem_triangles)
    assert (after[untouched] == before[untouched]).all()
    ctx.close()


def test_update_scene_argument_errors(etx, golden_dir):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_gems_128.etxscene"))
    ctx = etx.api.Context(0)
    with pytest.raises(etx.EtxHipError, match="no scene uploaded"):
        ctx.update_scene(snap, etx.api.CHANGED_CAMERA)
    ctx.upload_scene(snap)
    with pytest.raises(etx.EtxHipError, match="unknown bits"):
        ctx.update_scene(snap, 64)
    other = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_classic_128.etxscene"))
    with pytest.raises(etx.EtxHipError, match="counts"):
        ctx.update_scene(other, etx.api.CHANGED_POSITIONS)
    with pytest.raises(etx.EtxHipError):  # the failed update left no scene
        ctx.trace_rays(make_rays(16, 1))
    ctx.upload_scene(snap)
    bigger = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_gems_1080p.etxscene"))
    with pytest.raises(etx.EtxHipError, match="film size"):
        ctx.update_scene(bigger, etx.api.CHANGED_CAMERA)
    ctx.close()


def replicate_gems(etx, golden_dir, copies, seed=9):
    return synthetic_scenes.replicate_gems(etx, os.path.join(golden_dir, "cornell_gems_128.etxscene"), copies, seed)


def test_device_built_tree_is_the_emulated_one_and_finds_the_sah_hits(etx, golden_dir):
    """etx_hip_set_bvh_builder(ETX_HIP_BVH_DEVICE_LBVH): the kernels of kernels_bvh_build.hip build the tree the host emulation of the
    same per-element functions builds (tests/test_host_lbvh.py checks that one without a GPU) - same node count, depth and stack
    bound - and its closest hits are those of the binned-SAH tree."""
    snap = replicate_gems(etx, golden_dir, 40)
    assert snap.triangle_count > 100000
    rc, emulated = etx.api.host_check_bvh(snap, builder=etx.api.BVH_DEVICE_LBVH)
    assert rc == 0
    linear = etx.api.Context(0)
    linear.set_bvh_builder(etx.api.BVH_DEVICE_LBVH)
    linear.upload_scene(snap)
    built = linear.bvh_info()
    assert (built["nodes"], built["depth"], built["stack_need"], built["triangles"]) == (emulated["nodes"], emulated["depth"], emulated["stack_need"], emulated["triangles"])
    sah = etx.api.Context(0)
    sah.upload_scene(snap)
    reference = sah.bvh_info()
    rays = make_rays(200000, 31)
    hits_l, hits_s = linear.trace_rays(rays), sah.trace_rays(rays)
    same = hit_triangles(hits_l) == hit_triangles(hits_s)
    assert same.mean() > 0.9995 and (hit_triangles(hits_s) >= 0).mean() > 0.4
    np.testing.assert_allclose(hits_l[same, 2], hits_s[same, 2], rtol=1e-6, atol=1e-6)
    print("tree build, %d triangles: device linear %.2f ms (%d nodes, depth %d, stack %d), host binned SAH %.1f ms (%d nodes, depth %d, stack %d)" % (
        snap.triangle_count, built["build_ms"], built["nodes"], built["depth"], built["stack_need"], reference["build_ms"], reference["nodes"], reference["depth"], reference["stack_need"]))
    assert built["build_ms"] < reference["build_ms"]
    # the renders agree: same seeds, same hits
    films = []
    for ctx_builder in (etx.api.BVH_DEVICE_LBVH, etx.api.BVH_HOST_SAH):
        snap.samples = 4
        integ = etx.HIPPathTracing(snap)
        integ.context.set_bvh_builder(ctx_builder)
        integ.options()["bn"] = False
        z = np.load(os.path.join(golden_dir, "cie_observer.npz"))
        integ.cie_table = (z["xyz"], float(z["first_wavelength"]))
        films.append(render(etx, integ))
        integ.context.close()
    assert_same_render(films[0], films[1], "linear tree vs SAH tree")
    linear.close()
    sah.close()


def test_rebuild_on_the_device_after_moved_vertices(etx, golden_dir):
    """ETX_HIP_CHANGED_POSITIONS | ETX_HIP_REBUILD_BVH: a new tree over the moved vertices instead of the refit."""
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_gems_128.etxscene"))
    rays = make_rays(60000, 12)
    ctx = etx.api.Context(0)
    ctx.upload_scene(snap)  # host tree first
    before = ctx.bvh_info()
    move_material(snap, gem_material(snap), (0.11, 0.07, -0.05))
    ctx.update_scene(snap, etx.api.CHANGED_POSITIONS | etx.api.REBUILD_BVH)
    after = ctx.bvh_info()
    assert after["triangles"] == before["triangles"] and after["nodes"] != before["nodes"]  # a different builder's tree
    rebuilt = ctx.trace_rays(rays)
    fresh = etx.api.Context(0)
    fresh.upload_scene(snap)
    reference = fresh.trace_rays(rays)
    same = hit_triangles(rebuilt) == hit_triangles(reference)
    assert same.mean() > 0.9995
    np.testing.assert_allclose(rebuilt[same, 2], reference[same, 2], rtol=1e-6, atol=1e-6)
    # and a refit of the rebuilt tree works like a refit of the host's
    move_material(snap, gem_material(snap), (-0.05, 0.0, 0.02))
    ctx.update_scene(snap, etx.api.CHANGED_POSITIONS)
    fresh.upload_scene(snap)
    refit, reference = ctx.trace_rays(rays), fresh.trace_rays(rays)
    same = hit_triangles(refit) == hit_triangles(reference)
    assert same.mean() > 0.9995
    with pytest.raises(etx.EtxHipError, match="REBUILD_BVH"):
        ctx.update_scene(snap, etx.api.REBUILD_BVH)
    ctx.close()
    fresh.close()
