"""Diagnostic: where do non-finite camera-layer values come from (per max path length / option)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import etx_tracer_amd as etx
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_rough_128"
gd = os.path.join(ROOT, "tests", "golden")
out = {}
def vcm(name, spp, mpl, opts):
    snap = etx.SceneSnapshot(os.path.join(gd, scene + ".etxscene")); snap.samples = spp
    if mpl: snap.max_path_length = mpl
    integ = etx.HIPVCM(snap)
    integ.options()["vcm-blue_noise"] = False
    integ.options().update(opts)
    integ.render()
    cam = integ.film(etx.api.LAYER_CAMERA)[..., :3].copy(); lig = integ.film(etx.api.LAYER_LIGHT)[..., :3].copy(); integ.context.close()
    bad = ~np.isfinite(cam)
    print("max rgb", np.nan_to_num(cam).max(axis=(0,1)), "pixels with red marker", (cam[...,0] > 500).sum(), "green marker", (cam[...,1] > 500).sum())
    print("%-14s mpl %4d  cam nonfinite per channel %s  neg %s  light nonfinite %s  cam mean (finite) %s" % (
        name, mpl, bad.sum(axis=(0, 1)), (np.nan_to_num(cam) < 0).sum(axis=(0, 1)), (~np.isfinite(lig)).sum(axis=(0, 1)), np.nan_to_num(cam).mean(axis=(0, 1))))
    out[name + "_%d" % mpl] = cam
only_nee = {"vcm-merging": False, "vcm-connect_vertices": False, "vcm-connect_to_camera": False, "vcm-direct_hit": False}
for rep in range(3):
    vcm("only_nee", 64, 2, only_nee)
    vcm("all", 64, 3, {})
    vcm("all_nomerge", 64, 3, {"vcm-merging": False})
