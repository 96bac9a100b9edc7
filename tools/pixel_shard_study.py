"""configs[3] (sssdragon_bdpt, 1920x1080, BDPTFull) under pixel-interleaved sharding on ONE device: an unsharded context, then two contexts that
render pixels 0, 2, ... and 1, 3, ... (etx_hip_begin_ex) - working sets, throughput of each context alone, and the difference between the sum
of the two films and the unsharded film. What a two-GPU job would hold per GPU (no second GPU is available to this work: the contexts run one
after the other).   python tools/pixel_shard_study.py [iterations]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import etx_tracer_amd as etx
    from etx_tracer_amd import api, integrator as integ_mod
    from tools import synthetic_scenes, bluenoise_tables
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    snap = synthetic_scenes.sss_dragon(etx, os.path.join(ROOT, "tests", "golden", "cornell_sss_1080p.etxscene"))
    width, height = snap.film_size
    options = integ_mod.bdpt_options_from_dict({"bdpt-mode": api.BDPT_MODE_FULL})
    table = bluenoise_tables.load(os.path.join(ROOT, "tests", "golden", "bluenoise_64spp.npz"))
    rows, films = [], {}
    for name, first, stride in (("unsharded", 0, 1), ("unsharded, second run", 0, 1), ("pixels 0 mod 2", 0, 2), ("pixels 1 mod 2", 1, 2)):
        ctx = api.Context(0)
        ctx.upload_scene(snap)
        ctx.upload_bluenoise(6, table)
        for timed in (False, True):  # the first run finds the pool sizes
            ctx.begin_bdpt(options, 0, 1, first, stride)
            t0 = time.perf_counter()
            for _ in range(iterations):
                ctx.render_iteration()
            ctx.sync()
            seconds = time.perf_counter() - t0
        stats = ctx.stats()
        films[name] = (ctx.read_film(api.LAYER_CAMERA)[..., :3], ctx.read_film(api.LAYER_LIGHT)[..., :3])
        rows.append({"context": name, "working_set_gb": round(ctx.device_bytes() / 1.0e9, 2), "iterations": iterations, "seconds": round(seconds, 3),
                     "msamples_per_s": round(width * height * iterations / stride / seconds / 1.0e6, 2), "pool_grows": int(stats.pool_grows),
                     "rays_extension": int(stats.rays_extension), "light_vertices": int(stats.light_vertices)})
        ctx.close()
    whole = films["unsharded"]
    total = [films["pixels 0 mod 2"][k] + films["pixels 1 mod 2"][k] for k in (0, 1)]
    report = {"workload": "sssdragon_bdpt %dx%d BDPTFull, blue noise on" % (width, height), "contexts": rows,
              "rays_sum_equals_unsharded": rows[2]["rays_extension"] + rows[3]["rays_extension"] == rows[0]["rays_extension"],
              "mean_camera": float(whole[0].mean()), "mean_light": float(whole[1].mean())}
    again = films["unsharded, second run"]
    for k, layer in enumerate(("camera", "light", "camera, second unsharded run", "light, second unsharded run")):
        other = total[k] if k < 2 else again[k - 2]
        k = k % 2
        diff = np.abs(other - whole[k])
        at = np.unravel_index(int(diff.argmax()), diff.shape)
        # fp32 sums in another order: the difference scales with the pixel's value (a firefly of 1e5 moves by 1e-2)
        report[layer] = {"max_abs_difference": float(diff.max()), "unsharded_value_there": float(whole[k][at]), "max_value": float(whole[k].max()),
                         "max_relative_difference_above_1e-3": float((diff / np.maximum(np.abs(whole[k]), 1.0e-3)).max()),
                         "pixels_off_by_more_than_1e-4_relative": int((diff > 1.0e-4 * np.maximum(np.abs(whole[k]), 1.0e-3)).sum())}
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
