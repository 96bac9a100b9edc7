#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X VCM backend (BASELINE.json: Msamples/s, 1080p Cornell VCM).

One "step" = one full VCM iteration (light pass, photon grid, camera pass with connections + merge, film
accumulation) over the whole 1920x1080 frame = 2 073 600 samples. N GPUs: every rank renders `steps` iterations of
its own shard of the iteration sequence (rank r: r, r+N, ...; weak scaling) and the film is reduced over RCCL at the
end of every iteration (--reduce-every 1 = north_star; k: every k-th): asynchronous, out of place, from a snapshot, on a
communication stream of its own while the lanes keep rendering (csrc/host_reduce.h); the last reduce of the timed
region is the blocking one (etx_hip_reduce_film) that leaves every rank with the whole-job film. All reduces are
inside the timed region. Scene upload / BVH build happen before the timed region (inputs resident in HBM).
The timed region is repeated (each region: exactly --steps steps between barrier + synchronize on both sides) at least
three times and until the regions add up to 2.5 s (--repeats 0, the default; a 20-step region of the headline workload
lasts 0.4 s - too short for an outside observer to see the device busy); `value` / `ms_per_step` are the median
region's, `repeats` lists all of them.

No torch in this process: the collectives (film reduce, barrier, max over ranks) are RCCL inside libetx_hip.so, the 128-byte
ncclUniqueId travels through a file on the node (etx_tracer_amd/multi_gpu.py). Importing the torch wheel would map the ROCm
7.0.2 runtime it bundles next to (or, imported first, instead of) the one the library is built against; `config.runtime`
records what the process runs on.

  python bench.py --gpus 1 --steps 8 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (traversal kernel, HBM) and
`cpu_baseline` (the reference's CPUVCM compiled into oracle/_ref, timed on this box' host cores on a bounded sample).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
BYTES_PER_RAY = 52         # 32 B ray record + 4 B path / queue index in, 16 B hit record out. (The queues here are index-aligned with the path state,
                           # so the kernel moves 48 of them - the index is implicit; the contract's figure is the one priced.)


def cpu_baseline(snapshot_path, width, height, seconds_budget=25.0, integrator="vcm", extra=()):
    """Times the reference's CPUVCM / CPUBidirectional (oracle/_ref/etx_oracle = reference integrator + BVH shim, no Embree) on
    all host cores, on a bounded number of iterations of the SAME workload."""
    binary = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")
    if not os.path.exists(binary):
        return None
    # one probe iteration, then as many as fit the budget
    def run(iterations):
        # VCMOptions::default_values() (blue noise on), like the device run; --max-iterations bounds the sample, --spp keeps
        # scene.samples (and with it the blue-noise class) at the workload's 64
        out = subprocess.run([binary, "--load-snapshot", snapshot_path, "--integrator", integrator, "--spp", "64", "--max-iterations", str(iterations), *extra],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        m = re.search(r"ORACLE_RESULT (\{.*\})", out.stdout)
        return json.loads(m.group(1)) if m else None

    probe = run(1)
    if probe is None:
        return None
    iterations = max(1, min(16, int(seconds_budget / max(probe["seconds"], 1e-3))))
    result = run(iterations) if iterations > 1 else probe
    if result is None:
        return None
    return {
        "value": round(result["msamples_per_s"], 4),
        "unit": "Msamples/s",
        "cores": os.cpu_count(),
        "threads": result["threads"],
        "kind": "reference",
        "sample": "%d %s iterations of the same %dx%d snapshot (reference %s + oracle BVH shim instead of Embree), %.1f s" % (
            result["iterations"], integrator.upper(), width, height, "CPUVCM" if integrator == "vcm" else "CPUBidirectional", result["seconds"]),
    }


def flush_c_stdio():
    """RCCL (and HIP) write diagnostics through C stdio, which is block-buffered when stdout is a pipe and flushed at exit - after Python's own output."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def library_sha16():
    import hashlib
    from etx_tracer_amd import api
    try:
        with open(api.library_path(), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def main(argv=None, context_factory=None):
    """`context_factory`: the CPU test of the N > 1 control flow (tests/test_multi_gpu_gloo.py) runs this function in two processes with a
    stand-in for api.Context (whose collectives run over gloo); the driver's command line does not use it."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=32)
    parser.add_argument("--warmup", type=int, default=8)
    parser.add_argument("--workload", default="full", choices=["full", "classic", "gems", "gems1m", "sssdragon_bdpt", "cloud_bdpt"],
                        help="full = surviving cornellbox.mtl (fog medium, env + dir emitters) = BASELINE.json configs[1]; classic = area light only; "
                             "gems = configs[2] family: 2 892 triangles (BVH4 traversal), dispersive dielectrics + rough conductor, spectral; "
                             "gems1m = the same with 350 scaled copies of its gems (1 010 892 triangles: mesh size of configs[3-4], tools/synthetic_scenes.py); "
                             "sssdragon_bdpt = configs[3]: two random-walk subsurface blob meshes of 102 400 triangles, BDPTFull, 1920x1080; "
                             "cloud_bdpt = configs[4]: the fog box with a procedural 256^3 heterogeneous density grid, BDPTFull, 2048x2048")
    parser.add_argument("--bvh", default="host", choices=["host", "device"], help="who builds the traversal tree (etx_hip_set_bvh_builder)")
    parser.add_argument("--shard", default="iterations", choices=["iterations", "pixels"],
                        help="how N > 1 ranks split the job. iterations (default): rank r renders iterations r, r + N, ... of the full frame - weak scaling, every integrator. "
                             "pixels: every rank renders ALL iterations of pixels r, r + N, ... (etx_hip_begin_ex; bidirectional workloads only) - strong scaling: the job "
                             "is --steps iterations of the frame whatever N is, and each rank holds 1/N of the vertex pools")
    parser.add_argument("--reduce-every", type=int, default=1,
                        help="film reduce cadence in iterations (N > 1, or N = 1 with --comm-single): 1 = at the end of every iteration (north_star), k = every k-th, "
                             "0 = only the final reduce of the timed region")
    parser.add_argument("--repeats", type=int, default=0, help="timed regions of --steps steps each; value = the median one. 0 (default): at least 3, and until the regions add up to 2.5 s (at most 24)")
    parser.add_argument("--comm-single", action="store_true",
                        help="N = 1 only: create a one-rank RCCL communicator, so that the reduces of the timed region run (snapshot kernel + one-rank all-reduce) and their "
                             "device time can be read on one GPU (reduce.device_ms_avg); without it a single GPU has nothing to reduce")
    parser.add_argument("--debug-flags", type=int, default=0, help="etx_hip_set_debug_flags before the scene upload: kernel variants for A/B runs (64: two-ray packed sweep, 128: matrix-core sweep); 0 = the product")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-kernel-table", action="store_true", help="skip the extra pass that times every kernel group (profiling runs)")
    args = parser.parse_args(argv)

    import numpy as np
    import etx_tracer_amd as etx
    from etx_tracer_amd import api, multi_gpu, integrator as integ_mod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus))
    distributed = world > 1
    if distributed and (os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1")):
        # one node, rendezvous on the loopback address (the launch contract): RCCL's bootstrap sockets stay on it as well instead of the first interface
        # it finds (a container hostname that does not resolve must not matter); the film itself travels over xGMI / shared memory, not sockets
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

    spectral_workload = args.workload in ("gems", "gems1m")
    bdpt_workload = args.workload in ("sssdragon_bdpt", "cloud_bdpt")
    snapshot_path = os.path.join(ROOT, "tests", "golden", "cornell_%s_1080p.etxscene" % ("gems" if spectral_workload else args.workload))
    cpu_snapshot_path, cpu_extra = snapshot_path, ()
    if args.workload == "gems1m":
        from tools import synthetic_scenes
        snap = synthetic_scenes.replicate_gems(etx, snapshot_path, 350)
    elif args.workload == "sssdragon_bdpt":
        from tools import synthetic_scenes
        snapshot_path = os.path.join(ROOT, "tests", "golden", "cornell_sss_1080p.etxscene")
        snap = synthetic_scenes.sss_dragon(etx, snapshot_path)
        cpu_snapshot_path = "/tmp/etx_bench_sssdragon_%d.etxscene" % os.getpid()  # the assembled scene, for the reference's driver
    elif args.workload == "cloud_bdpt":
        snapshot_path = os.path.join(ROOT, "tests", "golden", "cornell_cloud_2048.etxscene")
        snap = etx.SceneSnapshot(snapshot_path)
        snap.inject_density(256)
        cpu_snapshot_path, cpu_extra = snapshot_path, ("--inject-density", "256")  # the driver builds the same grid (oracle/driver/etx_oracle.cxx)
    else:
        snap = etx.SceneSnapshot(snapshot_path)
    width, height = snap.film_size
    ctx = (context_factory or api.Context)(local_rank)
    ctx.set_bvh_builder(api.BVH_DEVICE_LBVH if args.bvh == "device" else api.BVH_HOST_SAH)
    if args.debug_flags:
        ctx.set_debug_flags(args.debug_flags)  # kept by the pipelines etx_hip_upload_scene allocates
    lanes = ctx.lanes(api.INTEGRATOR_BDPT if bdpt_workload else api.INTEGRATOR_VCM)  # iterations in flight (etx_hip_lanes)
    upload_t0 = time.perf_counter()
    ctx.upload_scene(snap)
    upload_seconds = time.perf_counter() - upload_t0
    tree = ctx.bvh_info()
    if spectral_workload:  # spectral scene: the host's CIE observer (committed fixture of the reference's table)
        cie = np.load(os.path.join(ROOT, "tests", "golden", "cie_observer.npz"))
        ctx.upload_cie_table(cie["xyz"], float(cie["first_wavelength"]))
    if distributed:
        multi_gpu.init_context_comm(ctx, rank, world)
    elif args.comm_single:
        ctx.comm_init(0, 1, api.comm_unique_id(ctx.library))
    flush_c_stdio()  # the communicator's banner (every rank that printed one) goes out now, not behind the JSON line

    # VCMOptions::default_values(): blue noise on. The C++ host tabulates its BNSampler for the class of scene.samples
    # (64 -> set 6, include/etx_hip.h); here the same table comes from the committed fixture of the reference's sampler.
    from tools import bluenoise_tables
    snap_samples = snap.samples
    if bluenoise_tables.set_index(snap_samples) != 6:
        raise SystemExit("bench workload expects scene.samples = 64 (blue-noise class 6), snapshot has %d" % snap_samples)
    ctx.upload_bluenoise(6, bluenoise_tables.load(os.path.join(ROOT, "tests", "golden", "bluenoise_64spp.npz")))
    options = integ_mod.vcm_options_from_dict({})
    if bdpt_workload:  # CPUBidirectional's defaults except the mode: BDPTFull = all vertex connections
        options = integ_mod.bdpt_options_from_dict({"bdpt-mode": api.BDPT_MODE_FULL})
        ctx.set_timers(0xff)  # the roofline kernel of these workloads is whichever group dominates: time all of them

    pixel_sharded = (args.shard == "pixels") and (world > 1)
    if pixel_sharded and (bdpt_workload is False):
        raise SystemExit("--shard pixels: the bidirectional workloads only (a VCM iteration's photon map needs the light paths of every pixel)")

    def begin(first_iteration, iteration_stride):
        if pixel_sharded:
            ctx.begin_bdpt(options, first_iteration=first_iteration, iteration_stride=iteration_stride, pixel_first=rank, pixel_stride=world)
        elif bdpt_workload:
            ctx.begin_bdpt(options, first_iteration=first_iteration, iteration_stride=iteration_stride)
        else:
            ctx.begin_vcm(options, first_iteration=first_iteration, iteration_stride=iteration_stride)

    def run_steps(count, first_offset):
        if pixel_sharded:
            begin(first_offset, 1)  # the same iterations on every rank, each for its own pixels
        else:
            begin(rank + first_offset * world, world)
        reduces_before = ctx.reduce_info()
        host_blocked = 0.0
        for step in range(count):
            ctx.render_iteration()  # asynchronous: iterations overlap on the device lanes
            # north_star: the film reduce "at the end of each iteration". Enqueued behind the commits handed over so far, on the communication
            # stream; the lanes keep rendering. (The last one of the region is the blocking etx_hip_reduce_film below.)
            if (args.reduce_every > 0) and ((step + 1) % args.reduce_every == 0) and (step + 1 < count):
                t_reduce = time.perf_counter()
                ctx.reduce_film_begin()  # never waits: any number of reduces may be in flight, the communication stream orders them
                host_blocked += time.perf_counter() - t_reduce
        ctx.sync()
        s = ctx.stats()             # totals since begin
        acc = {"rays": s.rays_extension, "trace_ms": s.ms_trace_closest, "launches": s.launches_trace_closest, "shadow": s.rays_shadow, "lv": s.light_vertices,
               "rounds": s.wavefront_bounces, "examined": s.photons_examined, "stats": s}
        t_reduce = time.perf_counter()
        ctx.reduce_film()           # every iteration of every rank is in the reduced copy; rendering could go on
        final_blocked = time.perf_counter() - t_reduce
        after = ctx.reduce_info()
        n_reduces = int(after.reduces - reduces_before.reduces)
        acc["reduce"] = {
            "every": args.reduce_every, "count": n_reduces, "payload_mb": round(after.payload_bytes / 1.0e6, 2),
            # device time of a reduce = snapshot kernel + collectives, HIP events on the communication stream (overlaps the lanes' kernels)
            "device_ms_avg": round((after.total_device_ms - reduces_before.total_device_ms) / n_reduces, 4) if n_reduces else None,
            "device_ms_last": round(after.last_device_ms, 4) if n_reduces else None,
            # what the host thread spent inside the reduce calls: the asynchronous ones (waiting for a predecessor still in flight) and the final blocking one
            "host_blocked_async_ms": round(host_blocked * 1.0e3, 4), "host_blocked_final_ms": round(final_blocked * 1.0e3, 4),
        }
        return acc

    def barrier():
        ctx.sync()          # this rank's lanes are idle (hipStreamSynchronize of every launch stream: what torch.cuda.synchronize would wait for)
        ctx.comm_barrier()  # every rank is here (an RCCL all-reduce on the communication stream; nothing to do on one rank)

    if args.warmup > 0:
        run_steps(args.warmup, 0)
    regions = []
    repeat = 0
    while True:
        barrier()
        t0 = time.perf_counter()
        region_acc = run_steps(args.steps, args.warmup + repeat * args.steps)
        barrier()
        region_elapsed = multi_gpu.max_over_ranks(ctx, time.perf_counter() - t0)  # the same number on every rank: all pick the same region, all stop together
        regions.append((region_elapsed, repeat, region_acc))
        repeat += 1
        if (repeat >= args.repeats) if (args.repeats > 0) else ((repeat >= 3) and ((sum(r[0] for r in regions) >= 2.5) or (repeat >= 24))):
            break
    # the headline is the MEDIAN region (its own steps, time and counters); min / max show the spread of this box
    elapsed, median_repeat, acc = sorted(regions, key=lambda r: r[0])[(len(regions) - 1) // 2]

    result = ctx.read_film(api.LAYER_RESULT)
    finite = bool(np.isfinite(result).all())

    # The same kernel alone on the device (no other lane's kernels sharing the CUs): 2 M incoherent rays inside the box,
    # device-resident queues, HIP events on the launch stream (etx_hip_trace_rays_device).
    isolated = None
    if rank == 0:
        n_rays = width * height
        g = np.random.default_rng(1)
        rays = np.empty((n_rays, 8), dtype=np.float32)
        rays[:, 0] = g.random(n_rays, dtype=np.float32) * 1.9 - 0.95
        rays[:, 1] = g.random(n_rays, dtype=np.float32) * 1.85 + 0.05
        rays[:, 2] = g.random(n_rays, dtype=np.float32) * 1.9 - 0.95
        rays[:, 3] = 2.2889e-4
        d = g.standard_normal((n_rays, 3), dtype=np.float32)
        rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
        rays[:, 7] = 3.0e38
        ms = ctx.trace_rays_timed(rays, 20)  # uploaded once, then device-resident (etx_hip_trace_rays_timed)
        del rays, d
        gbs = n_rays * BYTES_PER_RAY / ms / 1.0e6
        isolated = {"rays_per_launch": n_rays, "avg_launch_ms": round(ms, 6), "achieved": round(gbs, 3), "frac": round(gbs / HBM_PEAK_GBS, 6),
                    "note": "the same kernel alone on the device, 20 launches over one queue of incoherent rays"}

    # Per-kernel-group rooflines: a short extra pass with every kernel group timed by HIP events on its launch stream
    # (etx_hip_set_timers; the main timed region above times only the two traversal groups). Algorithmic bytes per unit are
    # the figures of DESIGN.md 3; the units come from the device counters of the same pass.
    def kernel_table(s, steps, note):
        merge_vertices = s.camera_vertices
        if bdpt_workload:
            # bidirectional state = 116 B (84 B + the previous vertex' position / normal), records of dev_bdpt.h
            units = {
                "trace_closest": ("ray", s.rays_extension, float(BYTES_PER_RAY) * s.rays_extension, s.ms_trace_closest, "k_trace_closest_bvh: SURVEY 8(d) 52 B per ray (32 B ray + 4 B index in, 16 B hit out; + the tree: not cache resident at this size)"),
                "trace_shadow": ("segment", s.rays_shadow, 48.0 * s.rays_shadow + 12.0 * s.splats, s.ms_trace_shadow, "k_trace_shadow: 48 B request in, 12 B of film atomics per visible light splat"),
                "shade_light": ("path segment", s.rays_light, 248.0 * s.rays_light + 96.0 * s.light_vertices, s.ms_shade_light,
                                "k_bdpt_light_shade + k_bdpt_walk + k_bdpt_connect_camera: 116 B state + 16 B hit in, 116 B state out, 96 B per stored light vertex (every event of a subsurface walk is one)"),
                "shade_camera": ("path segment", s.rays_camera, 248.0 * s.rays_camera + 164.0 * s.camera_vertices, s.ms_shade_camera,
                                 "k_bdpt_camera_shade + k_bdpt_walk + k_bdpt_connect_light: 116 B state + 16 B hit in, 116 B out, 116 B vertex record + 48 B request per connectible vertex"),
                "connect": ("pair", s.pairs, 220.0 * s.pairs, s.ms_connect, "k_bdpt_expand_pairs + k_bdpt_connect_pairs: 8 B pair + 96 B light vertex + 68 B camera vertex + 48 B shadow request"),
            }
        else:
            units = {
                "trace_closest": ("ray", s.rays_extension, float(BYTES_PER_RAY) * s.rays_extension, s.ms_trace_closest, "k_trace_closest: SURVEY 8(d) 52 B per ray (32 B ray + 4 B index in, 16 B hit out)"),
                "trace_shadow": ("segment", s.rays_shadow, 48.0 * s.rays_shadow + 12.0 * s.splats, s.ms_trace_shadow, "k_trace_shadow: 48 B request in, 12 B of film atomics per visible light splat"),
                "shade_light": ("path segment", s.rays_light, 184.0 * s.rays_light + 96.0 * s.light_vertices, s.ms_shade_light,
                                "k_light_shade (+ tail): 84 B state + 16 B hit in, 84 B state out, 96 B per stored light vertex"),
                "shade_camera": ("path segment", s.rays_camera, 184.0 * s.rays_camera + 164.0 * s.camera_vertices, s.ms_shade_camera,
                                 "k_camera_shade (+ tail): 84 B state + 16 B hit in, 84 B out, 116 B vertex record + 48 B NEE request per connectible vertex"),
                "connect": ("pair", s.pairs, 220.0 * s.pairs, s.ms_connect, "k_expand_pairs + k_connect_pairs: 8 B pair + 96 B light vertex + 68 B camera vertex + 48 B shadow request"),
                "merge": ("photon examined", s.photons_examined, 16.0 * s.photons_examined + 48.0 * s.photons_merged + 64.0 * merge_vertices, s.ms_merge,
                          "k_merge_* (sort + k_merge_diffuse / generic): 16 B per photon examined, 48 B per photon accepted, 8 x 8 B cell ranges per vertex"),
                "grid_build": ("light vertex", s.light_vertices, 176.0 * s.light_vertices, s.ms_grid_build, "k_grid_*: 96 B vertex in + 80 B photon record out"),
            }
        total_ms = sum(u[3] for u in units.values()) + s.ms_generate
        table = {}
        for name, (unit, count, nbytes, ms, what) in units.items():
            gbs = (nbytes / 1.0e9) / (ms * 1.0e-3) if ms > 0 else 0.0
            table[name] = {"bound": "hbm", "unit": unit, "units_per_step": round(count / steps), "ms_per_step": round(ms / steps, 4), "share": round(ms / total_ms, 4) if total_ms > 0 else None,
                           "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes": what}
        table["note"] = note
        return table

    kernels = None
    if (rank == 0) and bdpt_workload:
        kernels = kernel_table(acc["stats"], args.steps, "all kernel groups timed in the timed region itself (HIP events on the launch streams, %d iterations in flight): times include each "
                               "launch's dispatch and the sharing of the CUs with the other lanes' kernels" % lanes)
    elif (rank == 0) and (args.no_kernel_table == False):
        ctx.set_timers(0xff)
        begin(0, 1)
        groups_steps = min(args.steps, 8)
        for _ in range(groups_steps):
            ctx.render_iteration()
        ctx.sync()
        s = ctx.stats()
        ctx.set_timers(0x3)
        kernels = kernel_table(s, groups_steps, "%d extra steps with all kernel groups timed (HIP events on the launch streams, %d iterations in flight): times include each launch's dispatch and the "
                               "sharing of the CUs with the other lanes' kernels" % (groups_steps, lanes))

    # Counters of the one-lane rocprofv3 --pmc passes of THIS workload (tools/profile_round.sh -> profiles/round*_pmc_<workload>_1lane_summary.json):
    # the newest committed summary that names this workload; a group's figures come from the rows whose kernel names match it and are divided
    # by the units of work of the profiled run itself; whatever is not in the file is null (tools/profile_lookup.py).
    from tools import profile_lookup
    pmc_path, pmc = profile_lookup.summary_for(args.workload)
    pmc_groups = profile_lookup.BDPT_GROUPS if bdpt_workload else profile_lookup.VCM_GROUPS
    pmc_source = os.path.relpath(pmc_path, ROOT) if pmc_path else None
    # The counters describe the library they were collected on (tools/profile_round.sh records its hash in _meta). A summary of another build
    # is not evidence about this one: the counter fields are nulled and the line says so instead of quoting them.
    profiled_sha = (pmc or {}).get("_meta", {}).get("library_sha16")
    counters_stale = bool(pmc is not None and context_factory is None and profiled_sha != library_sha16())
    if counters_stale:
        pmc = None
    if kernels is not None:
        for group, prefixes in pmc_groups.items():
            if group in kernels:
                row = profile_lookup.group_counters(pmc, prefixes, profile_lookup.GROUP_UNITS[group])
                kernels[group]["counters_1lane"] = row
                if row and row.get("ms_per_step") and kernels[group]["units_per_step"]:
                    # the group ALONE on the device (one lane: no other iteration's kernels share the CUs): the algorithmic bytes of this run's
                    # units over the kernel time of the profiled one-lane run, scaled by the units of that run
                    bytes_per_unit = kernels[group]["achieved"] * 1.0e9 * kernels[group]["ms_per_step"] * 1.0e-3 / kernels[group]["units_per_step"]
                    exclusive = bytes_per_unit * row["units_per_step_profiled"] / (row["ms_per_step"] * 1.0e-3) / 1.0e9 if row.get("units_per_step_profiled") else None
                    kernels[group]["exclusive_1lane"] = {"ms_per_step": row["ms_per_step"], "achieved": round(exclusive, 1) if exclusive else None,
                                                         "frac": round(exclusive / HBM_PEAK_GBS, 5) if exclusive else None}
        kernels["counters_1lane_source"] = pmc_source
        kernels["counters_stale"] = counters_stale

    dominant = None
    if kernels is not None:
        name = max((k for k in kernels if isinstance(kernels[k], dict)), key=lambda k: kernels[k]["share"] or 0.0)
        dominant = dict(kernels[name], group=name, note="the kernel group with the largest share of the device time of an iteration; `bound` = the HBM roofline the contract asks for. "
                        "The group is not bandwidth-bound: its waves wait on dependent gathers and divergent branches (counters_1lane; DESIGN.md 3)")
    if rank == 0:
        samples = float(width) * height * args.steps * (1 if pixel_sharded else world)
        value = samples / elapsed / 1.0e6
        achieved = (acc["rays"] * BYTES_PER_RAY / 1.0e9) / (acc["trace_ms"] * 1.0e-3) if acc["trace_ms"] > 0 else 0.0
        trace_pmc = profile_lookup.group_counters(pmc, pmc_groups["trace_closest"], "rays_extension")
        line = {
            "metric": "Msamples/s (pixels x spp / s), %s" % ("BDPT" if bdpt_workload else "VCM"),
            "value": round(value, 4),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1.0e3, 4),
            "repeats": {"count": len(regions), "values": [round(samples / r[0] / 1.0e6, 4) for r in regions], "min": round(samples / max(r[0] for r in regions) / 1.0e6, 4),
                        "median": round(value, 4), "max": round(samples / min(r[0] for r in regions) / 1.0e6, 4),
                        "note": "each value: exactly `steps` steps between barrier + synchronize; `value` is the median region, whose counters the line reports"},
            "reduce": acc["reduce"],
            "higher_is_better": True,
            "scaling": "strong" if pixel_sharded else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("cornell_%s_%dx%d" % (args.workload, width, height)) if bdpt_workload else ("cornell_%s_vcm_1920x1080" % args.workload),
                "scene": {"sssdragon_bdpt": "BASELINE configs[3]: the Cornell box of the reference's camera with two closed blob meshes (102 400 triangles, tools/synthetic_scenes.py sss_dragon) "
                                            "under the random-walk subsurface materials of scenes/cornell/cornell_sss.mtl",
                          "cloud_bdpt": "BASELINE configs[4]: the fog Cornell box (env + dir + area emitters) whose medium is a procedural 256^3 heterogeneous density grid "
                                        "(67 MB, SceneSnapshot.inject_density = etx_oracle --inject-density)"}.get(
                    args.workload, "Cornell box rebuilt for the reference's surviving camera/materials (scenes/make_scenes.py), loaded by the reference loader"),
                "integrator": ("BDPT, bdpt-mode BDPTFull, other CPUBidirectional defaults (blue noise on), scene.samples 64, max-path-length 1023, rr start 6, RGB" if bdpt_workload else
                               "VCM, VCMOptions::default_values() (blue noise on), scene.samples 64, max-path-length 1023, rr start 6, %s" % ("spectral" if spectral_workload else "RGB")),
                "triangles": int(snap.triangle_count), "tree": {"builder": args.bvh, "build_ms": round(tree["build_ms"], 3), "nodes": tree["nodes"], "depth": tree["depth"], "stack_need": tree["stack_need"],
                                                                "upload_s": round(upload_seconds, 3)},
                "samples_per_step": width * height,
                "parallelism": ("pixel-sharded x%d (rank r: pixels r, r + N, ... of every iteration), RCCL film all-reduce of zero-padded sums, " if pixel_sharded
                                else "iteration-sharded x%d, RCCL film all-reduce (out of place, from a snapshot, overlapped with rendering), ") % world +
                               ("every %d iteration(s) + the final one" % args.reduce_every if args.reduce_every > 0 else "once at the end of the timed region"),
                "working_set_gb": round(ctx.device_bytes() / 1.0e9, 2),  # queues, pools, grid and film of all lanes (etx_hip_device_bytes)
                "pool_grows": int(acc["stats"].pool_grows),  # iterations of the timed region that overflowed a pool and were rendered again (0 once the pools have their size)
                "lanes": lanes, "workload_key": args.workload, "library_sha16": library_sha16(), "debug_flags": args.debug_flags,
                # the ROCm runtime this process ran on: versions (hipRuntimeGetVersion / the HIP_VERSION the library was compiled against, the same for RCCL)
                # and the files they are mapped from - one of each in a healthy process
                "runtime": api.runtime_info() if context_factory is None else None,
            },
            "roofline": {
                "kernel": ("k_trace_closest_bvh (ray queue -> hit queue; BVH4, top levels staged in LDS, persistent workgroups)" if spectral_workload
                           else "k_trace_closest (ray queue -> hit queue; sweep over the <= 64 pre-transformed primitives of the box)"),
                "bound": "hbm",
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": (round(trace_pmc["hbm_bytes_per_unit"] * acc["rays"] / max(acc["launches"], 1)) if trace_pmc and trace_pmc["hbm_bytes_per_unit"] else None),
                "traffic_note": ("HBM bytes per average launch = PMC bytes per ray of the traversal kernel INSIDE a one-lane run of this pipeline (%s B: FETCH_SIZE x 2 + WRITE_SIZE of %s, %s) "
                                 "x rays per launch of the timed region; algorithmic 52 B per ray (SURVEY 8d). A query that crosses a medium boundary also rewrites its path's ray, medium and "
                                 "path distance (about 112 B of state per crossing, counters.boundary_crossings_per_sample)" % (trace_pmc["hbm_bytes_per_unit"], ", ".join(trace_pmc["kernels"]), pmc_source))
                                if trace_pmc and trace_pmc["hbm_bytes_per_unit"] else
                                ("the committed PMC summary %s was collected on library %s, this run loaded %s: counters withheld (counters_stale)" % (pmc_source, profiled_sha, library_sha16())
                                 if counters_stale else "no committed PMC summary of this workload names the traversal kernel of this build"),
                "bytes_per_ray": BYTES_PER_RAY,
                "rays": acc["rays"],
                "launches": acc["launches"],
                "avg_launch_ms": round(acc["trace_ms"] / max(acc["launches"], 1), 6),
                "note": "algorithmic bytes = rays x 52 B (SURVEY 8d) summed over the timed region / summed HIP-event time of the trace launches (rank 0); rays include the "
                        "queries the kernel runs beyond medium boundaries it crosses itself (counters.boundary_crossings_per_sample). "
                        "The timed region overlaps several iterations on separate streams (ETX_HIP_LANES), so a launch shares the CUs with "
                        "other kernels; `isolated` is the same kernel alone",
                "isolated": isolated,
            },
            "kernels": kernels,
            "dominant_kernel": dominant,
            "counters_stale": counters_stale,
            "counters_profiled_library_sha16": profiled_sha,
            "counters": {
                "rays_per_sample": round((acc["rays"] + acc["shadow"]) / (float(width) * height * args.steps), 3),
                "light_vertices_per_path": round(acc["lv"] / (float(width) * height * args.steps), 3),
                "wavefront_rounds_per_step": round(acc["rounds"] / args.steps, 2),
                "boundary_crossings_per_sample": round(acc["stats"].boundary_crossings / (float(width) * height * args.steps), 3),
                "photons_examined_per_sample": round(acc["examined"] / (float(width) * height * args.steps), 2),
                "finite": finite,
                # per step, the units the per-kernel figures are divided by (tools/profile_round.sh copies them into its PMC summary)
                "units_per_step": {k: round(getattr(acc["stats"], k) / args.steps) for k in ("rays_extension", "rays_shadow", "rays_light", "rays_camera", "pairs", "photons_examined",
                                                                                             "light_vertices", "camera_vertices", "boundary_crossings")},
            },
        }
        if bdpt_workload and (dominant is not None):
            # configs[3-4]: the roofline kernel is the group that dominates the step (HIP events of the timed region itself).
            # traffic = HBM bytes per step of that group's kernels from the separate rocprofv3 --pmc passes of this command on one lane
            # (tools/profile_round.sh: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; KiB -> bytes)
            group_pmc = (dominant.get("counters_1lane") or {})
            traffic = group_pmc.get("hbm_bytes_per_step")
            line["roofline"] = {"kernel": dominant["group"] + ": " + dominant["bytes"], "bound": "hbm", "achieved": dominant["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": dominant["frac"], "traffic": traffic, "algorithmic_bytes_per_step": round(dominant["achieved"] * 1.0e9 * dominant["ms_per_step"] * 1.0e-3),
                                "units_per_step": dominant["units_per_step"], "unit_of_work": dominant["unit"], "ms_per_step": dominant["ms_per_step"],
                                "share_of_step": dominant["share"],
                                "note": "algorithmic bytes of the dominant kernel group (DESIGN.md 3) / summed HIP-event time of its launches in the timed region, rank 0; "
                                        "traffic = HBM bytes per step of the group's kernels in the one-lane PMC passes (%s)" % pmc_source}
        # (gems1m is assembled in memory and has no file to hand to the reference's driver; the subsurface scene is written out for it)
        if args.no_cpu_baseline or (world > 1) or (args.workload == "gems1m"):
            line["cpu_baseline"] = None
        else:
            if cpu_snapshot_path != snapshot_path:
                snap.save(cpu_snapshot_path)
            line["cpu_baseline"] = cpu_baseline(cpu_snapshot_path, width, height, integrator="bdpt" if bdpt_workload else "vcm",
                                                extra=(tuple(cpu_extra) + ("--opt", "bdpt-mode=3")) if bdpt_workload else ())
            if cpu_snapshot_path != snapshot_path:
                os.remove(cpu_snapshot_path)
        # RCCL prints its version banner on C stdio when the first communicator is created; C stdio is flushed at exit, i.e. AFTER Python's
        # line - flush it now so that the JSON line is the last thing on stdout (the driver reads one JSON line)
        flush_c_stdio()
        print(json.dumps(line), flush=True)
    if distributed:
        ctx.comm_barrier()
    ctx.close()
    return line if rank == 0 else None


if __name__ == "__main__":
    main()
