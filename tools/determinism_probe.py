"""Renders a snapshot twice in two fresh contexts and reports how the films differ (same iterations, same seeds: only the order
of float additions should differ). Usage: python tools/determinism_probe.py [scene ...]   env ETX_HIP_LANES=1|4
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
etx = importlib.import_module("etx-tracer_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def render(scene, spp, integrator):
    snap = etx.SceneSnapshot(os.path.join(GOLDEN, "cornell_%s_128.etxscene" % scene))
    snap.samples = spp
    cls = {"vcm": etx.HIPVCM, "pt": etx.HIPPathTracing, "bdpt": etx.HIPBidirectional}[integrator]
    integ = cls(snap)
    integ.options().update({"vcm-blue_noise": False, "bn": False, "bdpt-blue_noise": False})
    z = np.load(os.path.join(GOLDEN, "cie_observer.npz"))
    integ.cie_table = (z["xyz"], float(z["first_wavelength"]))
    integ.render()
    out = [integ.film(layer)[..., :3].astype(np.float64) for layer in (etx.api.LAYER_CAMERA, etx.api.LAYER_LIGHT)]
    stats = integ.status()
    integ.context.close()
    return out, stats


for scene in (sys.argv[1:] or ["gems", "full"]):
    for integrator in os.environ.get("PROBE_INTEGRATORS", "vcm,pt").split(","):
        (a, sa), (b, sb) = render(scene, 16, integrator), render(scene, 16, integrator)
        for name, x, y in (("camera", a[0], b[0]), ("light", a[1], b[1])):
            scale = max(float(np.abs(x).mean()), 1e-9)
            off = np.abs(x - y) > 1e-3 * (np.maximum(np.abs(x), np.abs(y)) + scale)
            worst = np.unravel_index(np.argmax(np.abs(x - y)), x.shape)
            print("%s %s %s lanes=%s: mean %.8g vs %.8g, max abs diff %.4g at %s (values %.6g / %.6g), %d of %d values off; vertices %d/%d vs %d/%d" % (
                scene, integrator, name, os.environ.get("ETX_HIP_LANES", "default"), x.mean(), y.mean(), np.abs(x - y).max(), worst, x[worst], y[worst], int(off.sum()), off.size,
                sa.light_vertices, sa.camera_vertices, sb.light_vertices, sb.camera_vertices))
