"""Diagnostic for the ETX_HIP_TRAP build: decode what the NaN trap of vcm_connect_to_light reports."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import etx_tracer_amd as etx
scene = "cornell_rough_128"
gd = os.path.join(ROOT, "tests", "golden")
def vcm(spp, mpl, opts):
    snap = etx.SceneSnapshot(os.path.join(gd, scene + ".etxscene")); snap.samples = spp
    snap.max_path_length = mpl
    integ = etx.HIPVCM(snap)
    integ.options()["vcm-blue_noise"] = False
    integ.options().update(opts)
    integ.render()
    cam = integ.film(etx.api.LAYER_CAMERA)[..., :3].copy() * spp; integ.context.close()
    return cam
only_nee = {"vcm-merging": False, "vcm-connect_vertices": False, "vcm-connect_to_camera": False, "vcm-direct_hit": False}
mode = int(os.environ.get("ETX_HIP_TRAP", "0"))
for rep in range(3):
    cam = vcm(8, 2, only_nee)
    hit = cam[..., 0] >= 999.5
    print("mode", mode, "rep", rep, "trapped pixels", int(hit.sum()), "nonfinite", (~np.isfinite(cam)).sum(axis=(0, 1)))
    vals = cam[hit]
    c = collections.Counter((int(round(v[0] - 1000.0)), round(float(v[1]), 5), round(float(v[2]), 5)) for v in vals)
    for k, n in c.most_common(12):
        print("   code %d  g %g  b %g   x%d" % (k[0], k[1], k[2], n))
