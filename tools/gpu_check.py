#!/usr/bin/env python3
"""Development helper (GPU box): render a snapshot with the HIP backend and compare with an oracle film.
usage: gpu_check.py snapshot.etxscene spp [oracle_film.raw] [key=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import etx_tracer_amd as etx  # noqa: E402
from tools import film_io  # noqa: E402


def block_mean(img, b):
    h, w = img.shape[:2]
    return img[: h // b * b, : w // b * b, :3].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


def main():
    snap = etx.SceneSnapshot(sys.argv[1])
    spp = int(sys.argv[2])
    oracle = None
    options = {"vcm-blue_noise": False}
    for a in sys.argv[3:]:
        if "=" in a:
            k, v = a.split("=", 1)
            options[k] = {"true": True, "false": False}.get(v, v)
        else:
            oracle = film_io.read_film(a)
    snap.samples = spp
    integ = etx.HIPVCM(snap)
    integ.options().update(options)
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    cie = np.load(os.path.join(golden, "cie_observer.npz"))  # only consulted by spectral scenes
    integ.cie_table = (cie["xyz"], float(cie["first_wavelength"]))
    if options.get("vcm-blue_noise"):
        from tools import bluenoise_tables  # the committed table of the 64-spp class (scene.samples 33..64)
        integ.bluenoise_tables = {6: bluenoise_tables.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bluenoise_64spp.npz"))}
    t0 = time.time()
    integ.render()
    wall = time.time() - t0
    st = integ.status()
    w, h = snap.film_size
    print("rendered %dx%d x %d spp in %.3f s wall, device total %.3f s -> %.3f Msamples/s" % (w, h, spp, wall, st.total_time, w * h * spp / st.total_time / 1e6))
    per_iteration = {k: (v / max(1, st.completed_iterations) if (k.startswith("ms_") or k.startswith("rays_") or k in ("light_vertices", "photons_examined", "photons_merged", "splats", "wavefront_bounces", "launches_trace_closest", "launches_trace_shadow")) else v)
                     for k, v in st.as_dict().items()}
    print(per_iteration)  # totals since begin, divided by the iteration count
    cam = integ.film(etx.api.LAYER_CAMERA)
    light = integ.film(etx.api.LAYER_LIGHT)
    res = integ.film(etx.api.LAYER_RESULT)
    print("gpu    camera mean", cam[..., :3].mean(axis=(0, 1)), "light mean", light[..., :3].mean(axis=(0, 1)), "nan:", int(np.isnan(res).sum()))
    out = os.environ.get("ETX_CHECK_OUT")
    if out:
        film_io.write_film(out + ".raw", cam, light, spp, st.total_time)
        film_io.save_png(out + ".png", res)
    if oracle is not None:
        print("oracle camera mean", oracle["camera"][..., :3].mean(axis=(0, 1)), "light mean", oracle["light"][..., :3].mean(axis=(0, 1)))
        for name, a, b in (("camera", cam, oracle["camera"]), ("light", light, oracle["light"]), ("result", res, oracle["result"])):
            print("%-7s rmse %.5f  block8 rmse %.5f  block32 rmse %.5f  rel mean diff %s" % (
                name, film_io.rmse(a, b), film_io.rmse(block_mean(a, 8), block_mean(b, 8)), film_io.rmse(block_mean(a, 32), block_mean(b, 32)),
                np.round((a[..., :3].mean(axis=(0, 1)) - b[..., :3].mean(axis=(0, 1))) / np.maximum(b[..., :3].mean(axis=(0, 1)), 1e-8), 4)))


if __name__ == "__main__":
    main()
