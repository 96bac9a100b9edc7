#!/usr/bin/env python3
"""Fused rounds (csrc/kernels_shade.inl kFuse: the shade kernel sweeps for the next segment) against the split pipeline, on the GPU box:
the same iterations rendered both ways with a debug build of host_api.cpp (tools/build_variant.sh dbg "" host_api.cpp; ETX_HIP_FUSE_TRACE
and ETX_HIP_DEBUG_FLAGS are tuning knobs of debug builds; flag 0x800 = the vertex-major pair order of k_expand_pairs) - films equal up to the order of the float atomics, ray / vertex / crossing counters EXACTLY equal.
usage: ETX_HIP_LIBRARY=etx-tracer_amd/variants/libetx_hip_dbg.so python tools/fuse_check.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def render(flavour, fuse, flags=0):
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
import etx_tracer_amd as etx
snap = etx.SceneSnapshot(os.path.join(%r, "tests", "golden", "cornell_%s_128.etxscene"))
snap.samples = 24
integ = etx.HIPVCM(snap)
integ.options()["vcm-blue_noise"] = False
integ.render()
s = integ.status()
np.savez(sys.argv[1], camera=integ.film(etx.api.LAYER_CAMERA), light=integ.film(etx.api.LAYER_LIGHT),
         counters=np.array([s.rays_extension, s.rays_shadow, s.light_vertices, s.camera_vertices, s.boundary_crossings, s.rays_light, s.rays_camera, s.pairs, s.photons_merged], dtype=np.uint64))
""" % (ROOT, ROOT, flavour)
    out = "/tmp/fuse_%s_%d_%d.npz" % (flavour, fuse, flags)
    env = dict(os.environ, ETX_HIP_FUSE_TRACE=str(fuse), ETX_HIP_DEBUG_FLAGS=str(flags))
    subprocess.check_call([sys.executable, "-c", code, out], env=env)
    return np.load(out)


def main():
    ok = True
    for flavour, what in (("classic", "fuse"), ("full", "fuse"), ("full", "pair order")):
        # fused against split rounds (both with the default pair order); the vertex-major pair order of rounds 1-4 (debug flag 0x800) against the default
        a, b = (render(flavour, 0), render(flavour, 1)) if what == "fuse" else (render(flavour, 0, 0x800), render(flavour, 0, 0))
        print("---", flavour, what)
        same_counters = bool((a["counters"] == b["counters"]).all())
        print(flavour, "counters split", a["counters"].tolist(), "fused", b["counters"].tolist(), "equal" if same_counters else "DIFFERENT")
        for layer in ("camera", "light"):
            d = np.abs(a[layer][..., :3] - b[layer][..., :3])
            close = np.allclose(a[layer][..., :3], b[layer][..., :3], rtol=2e-4, atol=2e-5)
            print("  %s: max abs diff %.3e, mean %.5f vs %.5f -> %s" % (layer, d.max(), a[layer][..., :3].mean(), b[layer][..., :3].mean(), "ok" if close else "MISMATCH"))
            ok = ok and close
        ok = ok and same_counters
    print("FUSE_CHECK", "PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
