#!/usr/bin/env python3
"""Sample-count series of the reference's adaptive path tracer (test infrastructure, like everything under oracle/).

    python3 oracle/gen_golden_adaptive.py

CPUPathTracing with Scene::noise_threshold = 0.1 (the loader's default) on the classic box, 128 x 128, 256 spp, bn = false: after every
completed iteration the driver counts the pixels Film::active_pixel still reports (oracle/driver/etx_oracle.cxx --adaptive-log) - what
the next iteration samples. tests/golden/cornell_classic_128_pt_adaptive_counts.npz holds that series and its sum, the number the device's
etx_hip_stats_t::active_pixels is compared with (tests/test_gpu_parity.py test_pt_adaptive_sampling). The render is deterministic
(per-pixel seeds; the mask depends on the film only), two runs give the same series.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    log = "/tmp/adaptive_%d.log" % os.getpid()
    series = []
    for run in range(2):
        subprocess.check_call([ORACLE, "--load-snapshot", os.path.join(GOLDEN, "cornell_classic_128.etxscene"), "--integrator", "pt", "--spp", "256", "--opt", "bn=false",
                               "--adaptive-log", log], stdout=subprocess.DEVNULL)
        rows = np.loadtxt(log, dtype=np.int64)
        os.remove(log)
        assert (rows[:, 0] == np.arange(1, 257)).all()
        series.append(rows[:, 1])
    assert (series[0] == series[1]).all(), "the reference's adaptive render is expected to be deterministic"
    active_after = series[0].astype(np.uint32)
    pixels = 128 * 128
    sampled = pixels + int(active_after[:-1].sum())  # iteration 0 samples everything, iteration k what was active after k completed ones
    out = os.path.join(GOLDEN, "cornell_classic_128_pt_adaptive_counts.npz")
    np.savez_compressed(out, active_after=active_after, sampled_pixel_iterations=np.int64(sampled), pixels=np.int32(pixels), spp=np.int32(256), noise_threshold=np.float32(0.1))
    print("%s: %d sampled pixel-iterations = %.2f per pixel; first estimate leaves %d active, the last %d" % (out, sampled, sampled / pixels, active_after[32], active_after[-1]))


if __name__ == "__main__":
    sys.exit(main())
