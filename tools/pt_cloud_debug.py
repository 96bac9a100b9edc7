import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import importlib
etx = importlib.import_module("etx-tracer_amd")
G = "tests/golden"
ref = np.load(os.path.join(G, "hi", "cornell_cloud_128_pt_4096.npz"))["camera"]
snap = etx.SceneSnapshot(os.path.join(G, "cornell_cloud_128.etxscene")); snap.samples = 4096
res = {}
for name, opts in (("default", {"bn": False}), ("no_mis", {"bn": False, "mis": False}), ("no_nee", {"bn": False, "nee": False}), ("no_direct", {"bn": False, "direct": False})):
    integ = etx.HIPPathTracing(snap); integ.options().update(opts); integ.render()
    res[name] = integ.film(etx.api.LAYER_CAMERA)[..., :3]; integ.context.close()
def bm(x, b=8):
    h, w = x.shape[:2]; return x.reshape(h//b, b, w//b, b, 3).mean(axis=(1, 3))
d = bm(res["default"]) - bm(ref)
print("block-8 rmse", np.sqrt((d**2).mean()), "mean rel", (res["default"].mean(axis=(0,1)) - ref.mean(axis=(0,1))) / ref.mean(axis=(0,1)))
lum = d.mean(axis=2)
np.set_printoptions(linewidth=250, precision=1, suppress=True)
print((lum * 1e3).round(1))
print("ref block lum x1e3"); print((bm(ref).mean(axis=2) * 1e2).round(0))
np.savez_compressed("gpurun_out/pt_cloud_debug.npz", ref=ref, **res)
