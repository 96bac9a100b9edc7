import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import importlib
etx = importlib.import_module("etx-tracer_amd")

G = "tests/golden"
lo = np.load(os.path.join(G, "cornell_gems_128_vcm.npz")); hi = np.load(os.path.join(G, "hi", "cornell_gems_128_vcm_4096.npz")); hk = np.load(os.path.join(G, "hi", "cornell_gems_128_vcm_4096_rekeyed.npz"))
def tot(g): 
    t = g["camera"][..., :3] + g["light"][..., :3]; return np.where(np.isfinite(t), t, 0)
print("golden 64spp mean", tot(lo).mean(axis=(0,1)), "spp", lo["spp"]); print("golden 4096 mean", tot(hi).mean(axis=(0,1))); print("golden 4096 rekeyed", tot(hk).mean(axis=(0,1)))
import ctypes
cie = np.load(os.path.join(G, "cie_observer.npz")) if os.path.exists(os.path.join(G, "cie_observer.npz")) else None
print("cie", cie.files if cie is not None else None)
for spp, first in ((64, 0), (256, 0), (256, 256), (256, 2048)):
    snap = etx.SceneSnapshot(os.path.join(G, "cornell_gems_128.etxscene")); snap.samples = first + spp
    integ = etx.HIPVCM(snap, first_iteration=first); integ.options()["vcm-blue_noise"] = False
    integ.cie_table = (cie["xyz"], float(cie["first_wavelength"])) if cie is not None else None
    integ.render()
    t = (integ.film(etx.api.LAYER_CAMERA) + integ.film(etx.api.LAYER_LIGHT))[..., :3]; integ.context.close()
    print("device spp %d first %d mean" % (spp, first), t.mean(axis=(0,1)))
