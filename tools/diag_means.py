"""Diagnostic: per-channel mean radiance of device PT / VCM variants vs the golden films of one scene."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import etx_tracer_amd as etx

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_rough_128"
gd = os.path.join(ROOT, "tests", "golden")

def fin(x):
    ok = np.isfinite(x).all(axis=2)
    return np.where(ok[..., None], x, 0.0)

gv = np.load(os.path.join(gd, scene + "_vcm.npz")); gp = np.load(os.path.join(gd, scene + "_pt.npz"))
ref_vcm = fin(np.maximum(gv["camera"] + gv["light"], 0.0)); ref_pt = fin(gp["camera"])
print("golden vcm mean", ref_vcm.mean(axis=(0, 1)), "spp", int(gv["spp"]))
print("golden pt  mean", ref_pt.mean(axis=(0, 1)), "spp", int(gp["spp"]))

def vcm(spp, opts=None, first=0):
    snap = etx.SceneSnapshot(os.path.join(gd, scene + ".etxscene")); snap.samples = spp
    integ = etx.HIPVCM(snap, first_iteration=first, iteration_stride=1)
    integ.options()["vcm-blue_noise"] = False
    integ.options().update(opts or {})
    integ.render()
    res = integ.film(etx.api.LAYER_RESULT)[..., :3].copy(); integ.context.close()
    return res

def pt(spp):
    snap = etx.SceneSnapshot(os.path.join(gd, scene + ".etxscene")); snap.samples = spp
    integ = etx.HIPPathTracing(snap); integ.options()["bn"] = False
    integ.render()
    res = integ.film(etx.api.LAYER_CAMERA)[..., :3].copy(); integ.context.close()
    return res

out = {}
out["pt1024"] = pt(1024); print("dev pt 1024   ", out["pt1024"].mean(axis=(0, 1)))
for name, spp, opts, first in (
    ("vcm64", 64, None, 0), ("vcm64b", 64, None, 64), ("vcm512", 512, None, 0),
    ("nomerge", 256, {"vcm-merging": False}, 0),
    ("noconnv", 256, {"vcm-connect_vertices": False}, 0),
    ("nolight", 256, {"vcm-connect_to_light": False}, 0),
    ("nocam", 256, {"vcm-connect_to_camera": False}, 0),
    ("nomis_dh", 256, {"vcm-merging": False, "vcm-connect_vertices": False, "vcm-connect_to_light": False, "vcm-connect_to_camera": False}, 0),
):
    out[name] = vcm(spp, opts, first)
    m = out[name].mean(axis=(0, 1))
    print("dev %-9s" % name, m, "rel to golden vcm", (m - ref_vcm.mean(axis=(0, 1))) / ref_vcm.mean(axis=(0, 1)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "diag_" + scene + ".npz"), **out)
