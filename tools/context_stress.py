#!/usr/bin/env python3
"""Two host threads of one process, each creating a context, rendering a small scene and destroying the context, over and over (what the GPU tests' concurrent
halves did when the interpreter crashed once in GPU call r5k). Run under `python -X faulthandler`: a crash prints which ctypes call each thread was in.
usage: python -X faulthandler tools/context_stress.py [seconds]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import etx_tracer_amd as etx  # noqa: E402

SCENES = ["classic", "full", "gems", "rough", "sss"]
deadline = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)
counts = [0, 0]
cie = np.load(os.path.join(ROOT, "tests", "golden", "cie_observer.npz"))


def worker(k):
    n = 0
    while time.time() < deadline:
        flavour = SCENES[(n + 2 * k) % len(SCENES)]
        snap = etx.SceneSnapshot(os.path.join(ROOT, "tests", "golden", "cornell_%s_128.etxscene" % flavour))
        snap.samples = 12
        cls = (etx.HIPVCM, etx.HIPBidirectional, etx.HIPPathTracing)[n % 3]
        integ = cls(snap, first_iteration=k, iteration_stride=2)
        integ.options().update({"vcm-blue_noise": False, "bdpt-blue_noise": False, "bn": False})
        if flavour == "gems":
            integ.cie_table = (cie["xyz"], float(cie["first_wavelength"]))
        integ.render()
        film = integ.film(etx.api.LAYER_CAMERA)
        assert np.isfinite(film).all()
        integ.context.close()
        n += 1
        counts[k] = n


threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print("context_stress: %d + %d renders on two threads, no crash" % tuple(counts))
