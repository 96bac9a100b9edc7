import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import importlib
etx = importlib.import_module("etx-tracer_amd")
G = "tests/golden"
def bm(x, b=8):
    h, w = x.shape[:2]; return x[..., :3].reshape(h//b, b, w//b, b, 3).mean(axis=(1, 3))
for flavour in (sys.argv[1:] or ["classic", "full", "cloud", "glass"]):
    ref = np.load(os.path.join(G, "hi", "cornell_%s_128_vcm_4096_rekeyed.npz" % flavour))
    total = ref["camera"] + ref["light"]
    for mode in (3, 2, 0, 1):
        snap = etx.SceneSnapshot(os.path.join(G, "cornell_%s_128.etxscene" % flavour)); snap.samples = 1024
        integ = etx.HIPBidirectional(snap); integ.options().update({"bdpt-mode": mode, "bdpt-blue_noise": False})
        try:
            integ.render()
        except Exception as e:
            print(flavour, mode, "ERROR", e); integ.context.close(); continue
        cam, light = integ.film(etx.api.LAYER_CAMERA)[..., :3], integ.film(etx.api.LAYER_LIGHT)[..., :3]
        st = integ.status(); integ.context.close()
        dev = cam + light
        ok = np.isfinite(total).all(axis=2)
        rm = (dev[ok].mean(axis=0) - total[ok].mean(axis=0)) / total[ok].mean(axis=0)
        d = bm(np.where(ok[..., None], dev, 0)) - bm(np.where(ok[..., None], total, 0))
        print("total", np.round(dev[ok].mean(axis=0), 5)); print("%-8s mode %d  rel mean %s  block8 rmse %.3e  cam mean %.4f light mean %.4f (ref cam %.4f light %.4f) nonfinite %d overflow %d time %.1fs" % (
            flavour, mode, np.round(rm, 4), np.sqrt((d**2).mean()), cam.mean(), light.mean(), ref["camera"].mean(), ref["light"].mean(), st.nonfinite_dropped, st.overflow_flags, st.total_time), flush=True)
