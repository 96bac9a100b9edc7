"""Diagnostic for the interpreter crash of GPUTEST_r05: does anything write into host memory that the process has
already given back? Renders the two interleaved halves of a 128 x 128 frame exactly like tests/test_gpu_bdpt.py::render_halves
(two contexts one after the other, films read back, contexts destroyed), then maps CANARY regions - anonymous mappings the kernel
places into the address ranges that were just unmapped (film buffers of numpy, pinned mirrors, thread stacks), filled with a
pattern - watches them for a while, and finally does what the suite died in (`from scipy.ndimage import median_filter`).

    python tools/stray_write_probe.py [--torch] [--integrator bdpt|vcm|pt] [--spp 4096] [--no-film] [--canaries 96] [--watch 1.5]
Prints one line: PROBE ok|STRAY ... ; exit code 0 / 3 (stray write seen) / 139 (the interpreter died)."""
import argparse
import mmap
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def runtime_maps():
    keep = ("libamdhip64", "libhsa-runtime64", "librccl", "libetx_hip")
    seen = []
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.split()[-1]
            if any(k in path for k in keep) and path not in seen:
                seen.append(path)
    return seen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--torch", action="store_true", help="import torch first (what collecting tests/ did up to round 5)")
    ap.add_argument("--integrator", default="bdpt")
    ap.add_argument("--flavour", default="classic")
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--no-film", action="store_true")
    ap.add_argument("--canaries", type=int, default=96)
    ap.add_argument("--canary-bytes", type=int, default=256 * 1024)
    ap.add_argument("--watch", type=float, default=1.5)
    ap.add_argument("--no-scipy", action="store_true")
    args = ap.parse_args()
    if args.torch:
        import torch  # noqa: F401
    import numpy as np
    import etx_tracer_amd as etx
    golden = os.path.join(ROOT, "tests", "golden")

    def half(first):
        snap = etx.SceneSnapshot(os.path.join(golden, "cornell_%s_128.etxscene" % args.flavour))
        snap.samples = args.spp
        if args.integrator == "bdpt":
            integ = etx.HIPBidirectional(snap, first_iteration=first, iteration_stride=2)
            integ.options().update({"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False})
        elif args.integrator == "vcm":
            integ = etx.HIPVCM(snap, first_iteration=first, iteration_stride=2)
            integ.options().update({"vcm-blue_noise": False})
        else:
            integ = etx.HIPPathTracing(snap, first_iteration=first, iteration_stride=2)
            integ.options().update({"bn": False})
        integ.render()
        films = None
        if not args.no_film:
            films = (integ.film(etx.api.LAYER_CAMERA), integ.film(etx.api.LAYER_LIGHT))
        stats = integ.status()
        integ.context.close()
        return films, stats.completed_iterations

    t0 = time.time()
    results = [half(0), half(1)]
    checksum = 0.0
    for films, done in results:
        assert done == args.spp // 2
        if films is not None:
            checksum += float(films[0].sum() + films[1].sum())
    del results, films
    t1 = time.time()
    pattern = b"\x5a" * args.canary_bytes
    canaries = []
    for _ in range(args.canaries):
        m = mmap.mmap(-1, args.canary_bytes)
        m.write(pattern)
        canaries.append(m)
    stray = []
    deadline = time.time() + args.watch
    while time.time() < deadline and not stray:
        for index, m in enumerate(canaries):
            if m[:] != pattern:
                data = np.frombuffer(m[:], dtype=np.uint8)
                bad = np.nonzero(data != 0x5A)[0]
                address = np.frombuffer(m, dtype=np.uint8).ctypes.data
                stray.append((index, hex(address), int(bad[0]), int(bad[-1]), int(bad.size), bytes(data[bad[0]:bad[0] + 64]).hex()))
        time.sleep(0.01)
    tag = "STRAY %s" % stray if stray else "ok"
    print("PROBE %s torch=%s integrator=%s render %.2f s checksum %.4f runtime %s" % (tag, args.torch, args.integrator, t1 - t0, checksum, runtime_maps()), flush=True)
    if not args.no_scipy:
        from scipy.ndimage import median_filter  # noqa: F401
        print("PROBE scipy imported", flush=True)
    return 3 if stray else 0


if __name__ == "__main__":
    sys.exit(main())
