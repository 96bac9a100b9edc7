#!/usr/bin/env python3
"""Config-size reference films (BASELINE.json configs[1] and configs[2] at their own 1920x1080): runs the reference's CPUVCM
through the prebuilt oracle binary on the many-core host of the GPU box and keeps 8 x 8 block means of the film.

    gpurun -- 'python3 oracle/gen_golden_1080p.py'        (no GPU work; ~6 min of a 256-thread host)
    mv gpurun_out/golden_1080p/*.npz tests/golden/

  cornell_full_1080p_vcm_64_blocks.npz   64 iterations, vcm-blue_noise=false, ETX_ORACLE_DECORRELATE=2 (independent light / camera
  cornell_gems_1080p_vcm_8_blocks.npz     8 iterations   streams - the estimator the device implements)
      camera, light: float32 [135, 240, 3] block means; spp; seconds; threads

The block means keep the fixture small (a 1080p float film is 25 MB); the test (tests/test_gpu_parity_size.py) reduces the device
film the same way. Needs nothing of /root/reference at run time: the binary and the snapshots travel with the repository.
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import film_io  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(ROOT, "gpurun_out", "golden_1080p")


def block_mean(img, b=8):
    h, w = img.shape[:2]
    return img[: h // b * b, : w // b * b, :3].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3)).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    for flavour, spp in (("full", 64), ("gems", 8)):
        film_path = "/tmp/golden_1080p.raw"
        cmd = [ORACLE, "--load-snapshot", os.path.join(GOLDEN, "cornell_%s_1080p.etxscene" % flavour), "--integrator", "vcm", "--spp", str(spp), "--out", film_path,
               "--opt", "vcm-blue_noise=false"]
        env = dict(os.environ)
        env["ETX_ORACLE_DECORRELATE"] = "2"
        t0 = time.time()
        print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, env=env)
        film = film_io.read_film(film_path)
        os.remove(film_path)
        cam, light = film["camera"][..., :3], film["light"][..., :3]
        finite = np.isfinite(cam).all(axis=2) & np.isfinite(light).all(axis=2)  # the release build lets an occasional NaN sample through
        cam, light = np.where(finite[..., None], cam, 0.0), np.where(finite[..., None], light, 0.0)
        out = os.path.join(OUT, "cornell_%s_1080p_vcm_%d_blocks.npz" % (flavour, spp))
        np.savez_compressed(out, camera=block_mean(cam), light=block_mean(light), spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]), threads=np.int32(film["threads"]),
                            nonfinite_pixels=np.int32((~finite).sum()))
        print("  -> %s (%.0f s, %d threads, %d non-finite pixels)" % (out, time.time() - t0, film["threads"], (~finite).sum()), flush=True)


if __name__ == "__main__":
    main()
