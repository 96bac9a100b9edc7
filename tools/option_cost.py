#!/usr/bin/env python3
"""Cost of the parts of an iteration, measured by switching them off: renders the bench workload with VCMOptions subsets
(one device lane, every kernel group timed with HIP events) and prints ms per group per iteration.
    ETX_HIP_LANES=1 python tools/option_cost.py [--workload full] [--steps 4]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="full")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import etx_tracer_amd as etx
    from etx_tracer_amd import api, integrator as integ_mod
    from tools import bluenoise_tables
    snap = etx.SceneSnapshot(os.path.join(ROOT, "tests", "golden", "cornell_%s_1080p.etxscene" % args.workload))
    ctx = api.Context(0)
    ctx.upload_scene(snap)
    ctx.upload_bluenoise(6, bluenoise_tables.load(os.path.join(ROOT, "tests", "golden", "bluenoise_64spp.npz")))
    ctx.set_timers(0xff)
    cases = [
        ("all", {}),
        ("no connect_to_light", {"vcm-connect_to_light": False}),
        ("no direct_hit", {"vcm-direct_hit": False}),
        ("no connect_vertices", {"vcm-connect_vertices": False}),
        ("no merge_vertices", {"vcm-merge_vertices": False}),
        ("no connect_to_camera", {"vcm-connect_to_camera": False}),
        ("no mis", {"vcm-mis": False}),
        ("camera walk only", {"vcm-connect_to_light": False, "vcm-direct_hit": False, "vcm-connect_vertices": False, "vcm-merge_vertices": False, "vcm-connect_to_camera": False}),
    ]
    print("%-22s %8s %8s %8s %8s %8s %8s %8s %8s" % ("case", "total", "trace", "shadow", "sh_light", "sh_cam", "connect", "merge", "grid"))
    for name, values in cases:
        options = integ_mod.vcm_options_from_dict(values)
        for rep in range(2):  # first repetition warms up
            ctx.begin_vcm(options, 0, 1)
            for _ in range(args.steps):
                ctx.render_iteration()
            ctx.sync()
            s = ctx.stats()
        n = float(args.steps)
        print("%-22s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f" % (name, s.total_time / n * 1e3, s.ms_trace_closest / n, s.ms_trace_shadow / n, s.ms_shade_light / n, s.ms_shade_camera / n,
                                                                       s.ms_connect / n, s.ms_merge / n, s.ms_grid_build / n), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
