#!/usr/bin/env python3
"""Traversal kernel micro benchmark (GPU box): the production closest-hit kernel ALONE on a device-resident ray queue (etx_hip_trace_rays_timed:
host rays uploaded once, one untimed + `repeat` timed launches, HIP events on the launch stream), for camera-like and incoherent rays. No torch.

    trace_bench.py snapshot.etxscene [n_rays] [repeat] [--flags 0,64,128] [--check]

--flags: debug-flag variants of the flat sweep to run side by side (0 = the product's VALU sweep, 64 = two rays per lane on packed fp32,
         128 = affine part on the matrix cores, kernels_trace.hip k_trace_closest_mfma)
--check: every variant's hits are compared with variant 0's: same triangle, t / u / v within 1e-4 (absolute, scene units ~1); rays whose two nearest
         candidates lie within 1e-5 of each other (an edge shared by two primitives) may resolve to either."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import etx_tracer_amd as etx  # noqa: E402
from etx_tracer_amd import api  # noqa: E402


def make_rays(kind, n, g):
    rays = np.empty((n, 8), dtype=np.float32)
    if kind == "primary":
        # camera-like rays: origin (0, 1, 3.82), directions through a 16:9 image plane (fov ~39.6 deg)
        x = g.random(n, dtype=np.float32) * 2 - 1
        y = g.random(n, dtype=np.float32) * 2 - 1
        d = np.stack([x * 0.36, y * 0.36 * 9 / 16, -np.ones_like(x)], axis=1)
        rays[:, 0:3] = (0.0, 1.0, 3.82)
    else:
        rays[:, 0] = g.random(n, dtype=np.float32) * 1.9 - 0.95
        rays[:, 1] = g.random(n, dtype=np.float32) * 1.85 + 0.05
        rays[:, 2] = g.random(n, dtype=np.float32) * 1.9 - 0.95
        d = g.standard_normal((n, 3), dtype=np.float32)
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3] = 2.2889e-4
    rays[:, 7] = 3.0e38
    return rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("snapshot")
    ap.add_argument("n_rays", nargs="?", type=int, default=1920 * 1080)
    ap.add_argument("repeat", nargs="?", type=int, default=20)
    ap.add_argument("--flags", default="0")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    variants = [int(f) for f in args.flags.split(",")]
    snap = etx.SceneSnapshot(args.snapshot)
    g = np.random.default_rng(1)
    rays = {kind: make_rays(kind, args.n_rays, g) for kind in ("primary", "incoherent")}
    reference = {}
    failures = 0
    for flags in variants:
        ctx = api.Context(0)
        ctx.set_debug_flags(flags)  # kept by the pipelines etx_hip_upload_scene allocates
        ctx.upload_scene(snap)
        if flags == variants[0]:
            print("tree:", ctx.bvh_info(), "runtime:", api.runtime_info()["hip_runtime"])
        for kind in ("primary", "incoherent"):
            n = args.n_rays
            ms, hits = ctx.trace_rays_timed(rays[kind], args.repeat, want_hits=True)
            tri = hits[:, 3].view(np.uint32)
            line = "flags %3d %-10s %d rays: %.4f ms/launch  %.2f Grays/s  %.1f GB/s (52 B/ray)  hit fraction %.4f" % (
                flags, kind, n, ms, n / ms / 1e6, n * 52 / ms / 1e6, float((tri != 0xFFFFFFFF).mean()))
            if args.check:
                if flags == variants[0]:
                    reference[kind] = hits.copy()
                else:
                    ref = reference[kind]
                    ref_tri = ref[:, 3].view(np.uint32)
                    same_tri = tri == ref_tri
                    close = np.abs(hits[:, :3] - ref[:, :3]).max(axis=1) < 1.0e-4
                    # a different triangle at the same distance: the ray runs through an edge two primitives share
                    edge = (~same_tri) & (np.abs(hits[:, 2] - ref[:, 2]) < 1.0e-4 * np.maximum(1.0, np.abs(ref[:, 2]))) & (tri != 0xFFFFFFFF) & (ref_tri != 0xFFFFFFFF)
                    bad = ~((same_tri & (close | (tri == 0xFFFFFFFF))) | edge)
                    line += "  | vs flags %d: %d differ (%d through shared edges), %d WRONG" % (variants[0], int((~same_tri).sum()), int(edge.sum()), int(bad.sum()))
                    if bad.any():
                        failures += 1
                        i = int(np.nonzero(bad)[0][0])
                        line += "  first: ray %d %s -> %s (tri %d) vs %s (tri %d)" % (i, rays[kind][i], hits[i, :3], tri[i], ref[i, :3], ref_tri[i])
            print(line, flush=True)
        ctx.close()
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
