#!/usr/bin/env python3
"""Config-size reference films (BASELINE.json configs[1..4] at their own size): runs the reference's CPUVCM / CPUBidirectional
through the prebuilt oracle binary on the many-core host of the GPU box and keeps 8 x 8 block means of the film.

    gpurun -- 'python3 oracle/gen_golden_1080p.py [name ...]'        (no GPU work; minutes of a 256-thread host)
    mv gpurun_out/golden_1080p/*.npz tests/golden/

  cornell_full_1080p_vcm_64_blocks[_asis].npz        configs[1]: 64 VCM iterations, vcm-blue_noise=false
  cornell_full_1080p_vcm_64_blocks_bluenoise[_asis].npz   configs[1] with VCMOptions::default_values() (blue noise on): job full_bn
  cornell_gems_1080p_vcm_32_blocks.npz               configs[2]: 32 VCM iterations (job gems32); ..._128_blocks.npz: 128 iterations (job gems128)
  cornell_sssdragon_1080p_bdpt3_16_blocks[_asis].npz configs[3]: 16 BDPTFull iterations of the scene tools/synthetic_scenes.py sss_dragon
                                                     assembles (written out with SceneSnapshot.save for the driver)
  cornell_cloud_2048_bdpt3_8_blocks[_asis].npz       configs[4]:  8 BDPTFull iterations, 256^3 density grid (--inject-density 256)
      camera, light: float32 [H/8, W/8, 3] block means; spp; seconds; threads
  default flavour: ETX_ORACLE_DECORRELATE=2 (independent light / camera streams - the estimator the device implements);
  `_asis`: the unmodified reference (light path i and camera path i share their seed).

The block means keep the fixtures small (a 1080p float film is 25 MB); the test (tests/test_gpu_parity_size.py) reduces the device
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

# name -> (snapshot flavour, integrator, iterations, extra driver arguments, output stem)
JOBS = {
    "full": ("full_1080p", "vcm", 64, ["--opt", "vcm-blue_noise=false"], "cornell_full_1080p_vcm_64_blocks"),
    "full_bn": ("full_1080p", "vcm", 64, [], "cornell_full_1080p_vcm_64_blocks_bluenoise"),  # VCMOptions defaults: what bench.py times
    "gems32": ("gems_1080p", "vcm", 32, ["--opt", "vcm-blue_noise=false"], "cornell_gems_1080p_vcm_32_blocks"),
    "gems128": ("gems_1080p", "vcm", 128, ["--opt", "vcm-blue_noise=false"], "cornell_gems_1080p_vcm_128_blocks"),  # round 5: 23 min on the container's 8 cores
    "sssdragon": ("sssdragon", "bdpt", 16, ["--opt", "bdpt-blue_noise=false", "--opt", "bdpt-mode=3"], "cornell_sssdragon_1080p_bdpt3_16_blocks"),
    "cloud": ("cloud_2048", "bdpt", 8, ["--opt", "bdpt-blue_noise=false", "--opt", "bdpt-mode=3", "--inject-density", "256"], "cornell_cloud_2048_bdpt3_8_blocks"),
}


def block_mean(img, b=8):
    h, w = img.shape[:2]
    return img[: h // b * b, : w // b * b, :3].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3)).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or ["full", "gems32"]
    for name in names:
        as_is = name.endswith("_asis")
        flavour, integrator, spp, extra, stem = JOBS[name[:-5] if as_is else name]
        snapshot = os.path.join(GOLDEN, "cornell_%s.etxscene" % flavour)
        if flavour == "sssdragon":  # assembled in memory, handed to the driver as a file
            import etx_tracer_amd as etx
            from tools import synthetic_scenes
            snapshot = "/tmp/golden_sssdragon.etxscene"
            synthetic_scenes.sss_dragon(etx, os.path.join(GOLDEN, "cornell_sss_1080p.etxscene")).save(snapshot)
        film_path = "/tmp/golden_1080p.raw"
        cmd = [ORACLE, "--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", film_path] + extra
        env = dict(os.environ)
        if as_is == False:
            env["ETX_ORACLE_DECORRELATE"] = "2"
        t0 = time.time()
        print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, env=env)
        film = film_io.read_film(film_path)
        os.remove(film_path)
        cam, light = film["camera"][..., :3], film["light"][..., :3]
        finite = np.isfinite(cam).all(axis=2) & np.isfinite(light).all(axis=2)  # the release build lets an occasional NaN sample through
        cam, light = np.where(finite[..., None], cam, 0.0), np.where(finite[..., None], light, 0.0)
        out = os.path.join(OUT, stem + ("_asis" if as_is else "") + ".npz")
        np.savez_compressed(out, camera=block_mean(cam), light=block_mean(light), spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]), threads=np.int32(film["threads"]),
                            nonfinite_pixels=np.int32((~finite).sum()))
        print("  -> %s (%.0f s, %d threads, %d non-finite pixels)" % (out, time.time() - t0, film["threads"], (~finite).sum()), flush=True)


if __name__ == "__main__":
    main()
