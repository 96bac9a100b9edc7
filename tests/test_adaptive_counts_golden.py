"""CPU: the sample-count series of the reference's adaptive path tracer (tests/golden/cornell_classic_128_pt_adaptive_counts.npz,
oracle/gen_golden_adaptive.py) has the shape Film::estimate_noise_levels gives it (film.cxx:233-330, path_tracing.cxx:91-99): every pixel
is sampled until the first estimate (sample index 32), the mask changes only after even sample indices, converged pixels can be re-activated by noisy neighbours, and
the sum is what the GPU test compares etx_hip_stats_t::active_pixels with."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_adaptive_series_follows_the_estimate_rule():
    z = np.load(os.path.join(GOLDEN, "cornell_classic_128_pt_adaptive_counts.npz"))
    active, pixels = z["active_after"].astype(np.int64), int(z["pixels"])
    assert active.shape == (256,) and pixels == 128 * 128 and abs(float(z["noise_threshold"]) - 0.1) < 1e-6
    # kMinSamples = 32: the first estimate runs after sample index 32, i.e. once 33 iterations are complete
    assert (active[:32] == pixels).all() and active[32] < pixels
    # estimates run after even sample indices only: the count after 2k + 1 completed iterations holds for two iterations
    assert (active[32:254:2] == active[33:255:2]).all()
    # a converged pixel is sampled again when a pixel within 5 of it fails a later estimate (the spread passes of estimate_noise_levels):
    # the count goes down overall, but not monotonically
    steps = np.diff(active)
    assert (steps > 0).sum() > 0 and steps.max() < 0.01 * pixels and active[-1] < active[40] < active[32]
    assert 0.8 * pixels < active[-1] < pixels
    sampled = pixels + int(active[:-1].sum())
    assert sampled == int(z["sampled_pixel_iterations"]) == 4012866
