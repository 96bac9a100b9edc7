#!/usr/bin/env python3
"""Traversal kernel micro benchmark (GPU box): k_trace_closest on device-resident ray / hit queues.
usage: trace_bench.py snapshot.etxscene [n_rays] [repeat]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import etx_tracer_amd as etx  # noqa: E402
from etx_tracer_amd import api  # noqa: E402


def make_rays(kind, n, generator):
    if kind == "primary":
        # camera-like rays: origin (0, 1, 3.82), directions through a 16:9 image plane (fov ~39.6 deg)
        x = torch.rand(n, generator=generator, device="cuda") * 2 - 1
        y = torch.rand(n, generator=generator, device="cuda") * 2 - 1
        d = torch.stack([x * 0.36, y * 0.36 * 9 / 16, -torch.ones_like(x)], dim=1)
        o = torch.tensor([0.0, 1.0, 3.82], device="cuda").expand(n, 3)
    else:
        o = torch.stack([torch.rand(n, generator=generator, device="cuda") * 1.9 - 0.95, torch.rand(n, generator=generator, device="cuda") * 1.85 + 0.05,
                         torch.rand(n, generator=generator, device="cuda") * 1.9 - 0.95], dim=1)
        d = torch.randn(n, 3, generator=generator, device="cuda")
    d = d / d.norm(dim=1, keepdim=True)
    ro = torch.cat([o, torch.full((n, 1), 2.2889e-4, device="cuda")], dim=1).contiguous()
    rd = torch.cat([d, torch.full((n, 1), 3.0e38, device="cuda")], dim=1).contiguous()
    return ro, rd


def main():
    snap = etx.SceneSnapshot(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1920 * 1080
    repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ctx = api.Context(0)
    ctx.upload_scene(snap)
    print("tree:", ctx.bvh_info())
    g = torch.Generator(device="cuda").manual_seed(1)
    for kind in ("primary", "incoherent"):
        ro, rd = make_rays(kind, n, g)
        hits = torch.empty((n, 4), device="cuda")
        torch.cuda.synchronize()
        ms = ctx.trace_rays_device(ro.data_ptr(), rd.data_ptr(), n, hits.data_ptr(), repeat)
        torch.cuda.synchronize()
        hit_fraction = float((hits[:, 3].view(torch.int32) != -1).float().mean())
        print("%-10s %d rays: %.4f ms/launch  %.2f Grays/s  %.1f GB/s (48 B/ray)  hit fraction %.3f" % (kind, n, ms, n / ms / 1e6, n * 48 / ms / 1e6, hit_fraction))
    ctx.close()


if __name__ == "__main__":
    main()
