#!/usr/bin/env python3
"""High-sample-count golden films from the reference-based oracle (oracle/_ref/etx_oracle), for the tight parity
tests of tests/test_gpu_parity_hi.py ("a high-spp oracle reference, e.g. 4096 spp at reduced resolution").

    python3 oracle/gen_golden_hi.py [--spp 4096] [--cores 0-3] [name ...]

  tests/golden/hi/cornell_<flavour>_128_<integrator>_<spp>.npz   camera / light layers (float16 pairs are NOT used:
      the films are float32, compressed), reference CPUVCM with vcm-blue_noise=false and CPUPathTracing with bn=false
  tests/golden/hi/cornell_full_128_vcm_<spp>_decorrelated.npz    the same with ETX_ORACLE_DECORRELATE=1 (the BVH shim
      shifts the shared stream of a pixel's light and camera path by ray-dependent amounts)
  tests/golden/hi/cornell_<flavour>_128_bdpt<mode>_<spp>[_rekeyed].npz  CPUBidirectional (--integrators bdpt --bdpt-modes 3,0,1)
  tests/golden/hi/cornell_<flavour>_128_vcm_<spp>_<far_first|random_child>.npz   (--integrators orders) the unmodified reference under
      ETX_ORACLE_BVH_ORDER = another child order of the BVH shim
  tests/golden/hi/cornell_<flavour>_128_vcm_<spp>_shared_first_vertex.npz   (--integrators firstvertex) ETX_ORACLE_DECORRELATE=3
  tests/golden/hi/cornell_<flavour>_128_vcm_<spp>_bluenoise[_rekeyed].npz   (--integrators bluenoise) VCMOptions defaults = blue noise on
  tests/golden/hi/cornell_<flavour>_128_vcm_<spp>_opaque_none.npz   (--integrators opaque_none) ETX_ORACLE_BVH_DRAWS=opaque_none, shared seeds
  tests/golden/hi/cornell_<flavour>_128_bdpt3_<spp>_opaque_none.npz   (--integrators bdpt_opaque_none) the same pin for CPUBidirectional, BDPTFull
  tests/golden/hi/cornell_<flavour>_128_vcm_<spp>_rekeyed.npz        ETX_ORACLE_DECORRELATE=2: the camera path re-keys its sampler at
      its first segment = independent light / camera streams, the estimator the device implements

The snapshots are the committed tests/golden/cornell_<flavour>_128.etxscene files (oracle/gen_golden.py writes them).
Needs /root/reference only through the prebuilt oracle binary. ~8 min per VCM scene on 8 cores.
"""
import argparse
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
HI = os.path.join(GOLDEN, "hi")

FLAVOURS = ["full", "rough", "glass", "gems", "cloud", "sss", "classic", "diamond", "spectral"]


def render(snapshot, integrator, spp, out_npz, cores, env_extra=None, extra=()):
    if os.path.exists(out_npz):
        print("have", out_npz)
        return
    film_path = "/tmp/golden_hi_%d.raw" % os.getpid()
    cmd = [ORACLE, "--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", film_path, *extra]
    if cores:
        cmd = ["taskset", "-c", cores] + cmd
    env = dict(os.environ)
    env.update(env_extra or {})
    t0 = time.time()
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, env=env)
    film = film_io.read_film(film_path)
    os.remove(film_path)
    layers = {"camera": film["camera"][..., :3].astype(np.float32), "spp": np.int32(film["spp"]), "seconds": np.float64(film["seconds"]), "threads": np.int32(film["threads"])}
    if integrator in ("vcm", "bdpt"):
        layers["light"] = film["light"][..., :3].astype(np.float32)
    os.makedirs(HI, exist_ok=True)
    np.savez_compressed(out_npz, **layers)
    print("  -> %s (%.0f s)" % (out_npz, time.time() - t0), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--cores", default="")
    ap.add_argument("--integrators", default="vcm,pt,rekeyed")
    ap.add_argument("--bdpt-modes", default="3")
    ap.add_argument("names", nargs="*", default=FLAVOURS)
    args = ap.parse_args()
    integrators = args.integrators.split(",")
    for flavour in args.names:
        snapshot = os.path.join(GOLDEN, "cornell_%s_128.etxscene" % flavour)
        variant = []
        if flavour == "sssmeshcb":  # the sssmesh snapshot with both subsurface materials switched to the Christensen-Burley class by the driver
            snapshot = os.path.join(GOLDEN, "cornell_sssmesh_128.etxscene")
            variant = ["--subsurface-class", "2"]
        if "vcm" in integrators:
            render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_%s_128_vcm_%d.npz" % (flavour, args.spp)), args.cores, extra=["--opt", "vcm-blue_noise=false"] + variant)
        if (flavour == "full") and ("vcm" in integrators):
            render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_full_128_vcm_%d_decorrelated.npz" % args.spp), args.cores, env_extra={"ETX_ORACLE_DECORRELATE": "1"},
                   extra=["--opt", "vcm-blue_noise=false"])
        if "orders" in integrators:
            # the unmodified estimator (shared light / camera streams) under two more traversal orders of the BVH shim: how far the
            # reference's own film moves when only the order candidates reach alpha_test_pass changes (Embree's order is unknown)
            for order in ("far_first", "random_child"):
                render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_%s_128_vcm_%d_%s.npz" % (flavour, args.spp, order)), args.cores, env_extra={"ETX_ORACLE_BVH_ORDER": order},
                       extra=["--opt", "vcm-blue_noise=false"] + variant)
        if "firstvertex" in integrators:
            # mode 3: shared streams through the first camera vertex, independent from the second segment on - where the reference's
            # light / camera correlation sits (not in the first vertex; this film equals the re-keyed one)
            render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_%s_128_vcm_%d_shared_first_vertex.npz" % (flavour, args.spp)), args.cores, env_extra={"ETX_ORACLE_DECORRELATE": "3"},
                   extra=["--opt", "vcm-blue_noise=false"] + variant)
        if "bluenoise" in integrators:
            # VCMOptions::default_values(): blue noise ON (vcm_shared.cxx:6-13; the override of the first camera vertex in the first
            # 256 iterations, vcm_shared.hxx:941-945,1018-1022) - the option set bench.py times. As is and re-keyed.
            render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_%s_128_vcm_%d_bluenoise.npz" % (flavour, args.spp)), args.cores, extra=variant)
            render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_%s_128_vcm_%d_bluenoise_rekeyed.npz" % (flavour, args.spp)), args.cores, env_extra={"ETX_ORACLE_DECORRELATE": "2"},
                   extra=variant)
        if "opaque_none" in integrators:
            # the UNMODIFIED integrator (shared seeds) with the candidate draws of always-opaque triangles taken off the path's stream
            # (ETX_ORACLE_BVH_DRAWS=opaque_none, oracle/shims/raytracing_bvh.cxx): a film that no longer depends on the traversal order
            render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_%s_128_vcm_%d_opaque_none.npz" % (flavour, args.spp)), args.cores, env_extra={"ETX_ORACLE_BVH_DRAWS": "opaque_none"},
                   extra=["--opt", "vcm-blue_noise=false"] + variant)
        if "bdpt_opaque_none" in integrators:
            # the same pin for CPUBidirectional (BDPTFull): shared light / camera seeds (bidirectional.cxx:377-380), candidate draws of opaque triangles off the stream
            render(snapshot, "bdpt", args.spp, os.path.join(HI, "cornell_%s_128_bdpt3_%d_opaque_none.npz" % (flavour, args.spp)), args.cores, env_extra={"ETX_ORACLE_BVH_DRAWS": "opaque_none"},
                   extra=["--opt", "bdpt-blue_noise=false", "--opt", "bdpt-mode=3"])
        if "rekeyed" in integrators:
            # mode 2: the camera sub path draws from a stream of its own from its first segment on (oracle/shims/raytracing_bvh.cxx)
            render(snapshot, "vcm", args.spp, os.path.join(HI, "cornell_%s_128_vcm_%d_rekeyed.npz" % (flavour, args.spp)), args.cores, env_extra={"ETX_ORACLE_DECORRELATE": "2"},
                   extra=["--opt", "vcm-blue_noise=false"] + variant)
        if "bdpt" in integrators:
            # CPUBidirectional, bdpt-mode 3 = BDPTFull (bidirectional.cxx:323-330); shared streams and re-keyed like VCM
            for mode in args.bdpt_modes.split(","):
                opts = ["--opt", "bdpt-blue_noise=false", "--opt", "bdpt-mode=%s" % mode]
                render(snapshot, "bdpt", args.spp, os.path.join(HI, "cornell_%s_128_bdpt%s_%d.npz" % (flavour, mode, args.spp)), args.cores, extra=opts)
                if mode != "0":  # PathTracing mode has no light path to correlate with
                    render(snapshot, "bdpt", args.spp, os.path.join(HI, "cornell_%s_128_bdpt%s_%d_rekeyed.npz" % (flavour, mode, args.spp)), args.cores,
                           env_extra={"ETX_ORACLE_DECORRELATE": "2"}, extra=opts)
        if ("bdpt-novc" in integrators) and (flavour == "sss"):
            # BDPTFull without vertex connections on the subsurface box (tests/test_gpu_bdpt.py)
            render(snapshot, "bdpt", args.spp, os.path.join(HI, "cornell_sss_128_bdpt3_%d_novc_rekeyed.npz" % args.spp), args.cores, env_extra={"ETX_ORACLE_DECORRELATE": "2"},
                   extra=["--opt", "bdpt-blue_noise=false", "--opt", "bdpt-mode=3", "--opt", "bdpt-conn_connect_vertices=false"])
        if "pt" in integrators:
            # --noise-threshold 0: every pixel gets all samples (the scenes carry Scene::noise_threshold = 0.1, with which
            # CPUPathTracing stops sampling converged pixels after 32 samples: a "4096-spp" film would hold ~100-spp noise)
            render(snapshot, "pt", args.spp, os.path.join(HI, "cornell_%s_128_pt_%d.npz" % (flavour, args.spp)), args.cores, extra=["--opt", "bn=false", "--noise-threshold", "0"] + variant)


if __name__ == "__main__":
    main()
