#!/usr/bin/env python3
"""Finds the committed counter summaries (profiles/round*_pmc_<workload>_1lane_summary.json, written by tools/profile_round.sh +
tools/pmc_aggregate.py) that bench.py quotes, and reduces them per kernel GROUP.

A summary is used only if it says which workload it profiled and how many units of work its run processed (`_meta`, written from the
bench line of the profiled command itself) - so an instruction count is never divided by another run's unit count - and a group's figures
come only from rows whose kernel names match the group's prefixes; what is not found is reported as None, never filled from an older
file. tests/test_profiles_current.py checks on the CPU that the kernels a summary names exist in the shipped libetx_hip.so.
"""
import glob
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")

# kernel-name prefixes of the groups bench.py times (host_api.cpp: which launches sit inside which timer)
# (k_trace_closest<true / _bvh<true: the pipeline's kernels - kFromCounter; the <false instantiations are bench.py's isolated launches over a fixed queue)
VCM_GROUPS = {
    "trace_closest": ("k_trace_closest<true", "k_trace_closest_bvh<true"),
    "trace_shadow": ("k_trace_shadow",),
    "shade_light": ("k_light_shade", "k_path_tail<false", "k_connect_endpoints<false"),
    "shade_camera": ("k_camera_shade", "k_path_tail<true", "k_connect_endpoints<true"),
    "connect": ("k_expand_pairs", "k_connect_pairs"),
    "merge": ("k_merge_",),
    "grid_build": ("k_grid_", "k_scan_"),
}
BDPT_GROUPS = {
    "trace_closest": ("k_trace_closest<true", "k_trace_closest_bvh<true"),
    "trace_shadow": ("k_trace_shadow",),
    "shade_light": ("k_bdpt_light_shade", "k_bdpt_walk_light", "k_bdpt_walk_exit_light", "k_bdpt_connect_camera"),
    "shade_camera": ("k_bdpt_camera_shade", "k_bdpt_walk_camera", "k_bdpt_walk_exit_camera", "k_bdpt_connect_light"),
    "connect": ("k_bdpt_expand_pairs", "k_expand_pairs", "k_bdpt_connect_pairs"),
}
# the unit of work of a group = which per-step count of the bench line (`counters.units_per_step`) it is divided by
GROUP_UNITS = {"trace_closest": "rays_extension", "trace_shadow": "rays_shadow", "shade_light": "rays_light", "shade_camera": "rays_camera", "connect": "pairs",
               "merge": "photons_examined", "grid_build": "light_vertices"}


def short_name(name):
    """the template name without its argument list (the same reduction tools/pmc_aggregate.py applies to rocprofv3's Kernel_Name)"""
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).replace("etxd::", "")


def library_kernels(path=None):
    """short names of the kernels in the shipped library (symbols of the embedded gfx950 code object)"""
    path = path or os.path.join(ROOT, "etx-tracer_amd", "libetx_hip.so")
    raw = subprocess.run(["strings", "-n", "8", path], capture_output=True, text=True, check=True).stdout.split("\n")
    mangled = sorted({s for s in raw if re.match(r"^_ZN4etxd\d+k_", s)})
    names = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.split("\n")
    return {short_name(n) for n in names if n}


def summaries():
    """[(round number, path, summary)] of every summary that carries a _meta block, newest round first"""
    found = []
    for path in glob.glob(os.path.join(PROFILES, "round*_pmc_*_1lane_summary.json")):
        m = re.match(r"round(\d+)_", os.path.basename(path))
        with open(path) as f:
            data = json.load(f)
        if m and isinstance(data.get("_meta"), dict):
            found.append((int(m.group(1)), path, data))
    return sorted(found, key=lambda t: -t[0])


def summary_for(workload):
    for _, path, data in summaries():
        if data["_meta"].get("workload") == workload:
            return path, data
    return None, None


def group_counters(summary, prefixes, unit_key):
    """What the one-lane PMC passes say about the kernels of one group; None when the summary holds none of them."""
    if summary is None:
        return None
    meta = summary["_meta"]
    rows = {name: row for name, row in summary.items() if (name != "_meta") and name.startswith(tuple(prefixes))}
    if not rows:
        return None
    iterations = max(1, int(meta.get("iterations", 1)))
    units = float(meta.get("units_per_step", {}).get(unit_key, 0.0))
    main_name, main = max(rows.items(), key=lambda kv: kv[1].get("duration_us_sum", 0.0))

    def total(key):
        values = [r[key] for r in rows.values() if key in r]
        return sum(values) if values else None

    def ratio(num, den):
        return round(num / den, 4) if (num is not None) and den else None

    wave_cycles = total("SQ_WAVE_CYCLES_sum")
    hits, misses = total("TCC_HIT_sum_sum"), total("TCC_MISS_sum_sum")
    valu = total("SQ_INSTS_VALU_sum")
    fetch, write = total("FETCH_SIZE_sum"), total("WRITE_SIZE_sum")
    hbm_bytes = (2.0 * fetch + write) * 1024.0 if (fetch is not None) and (write is not None) else None  # KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
    return {
        "kernel": main_name,
        "kernels": sorted(rows),
        "ms_per_step": round(sum(r.get("duration_us_sum", 0.0) for r in rows.values()) * 1.0e-3 / iterations, 4),
        "valu_lane_instructions_per_unit": round(valu * 64.0 / iterations / units, 1) if (valu is not None) and units else None,
        "waves_waiting_share": ratio(total("SQ_WAIT_ANY_sum"), wave_cycles),
        "waves_issuing_share": ratio(total("SQ_ACTIVE_INST_ANY_sum"), wave_cycles),
        "l2_hit_rate": ratio(hits, (hits or 0.0) + (misses or 0.0)),
        "occupancy_percent_mean": main.get("OccupancyPercent_mean"),
        "valu_busy_percent_mean": main.get("VALUBusy_mean"),
        "hbm_bytes_per_step": round(hbm_bytes / iterations) if hbm_bytes is not None else None,
        "hbm_bytes_per_unit": round(hbm_bytes / iterations / units, 2) if (hbm_bytes is not None) and units else None,
        "units_per_step_profiled": round(units) if units else None,
    }


if __name__ == "__main__":
    import sys
    for rnd, path, data in summaries():
        print(rnd, os.path.relpath(path, ROOT), data["_meta"])
    if len(sys.argv) > 1:
        path, data = summary_for(sys.argv[1])
        groups = BDPT_GROUPS if "bdpt" in sys.argv[1] else VCM_GROUPS
        for g, prefixes in groups.items():
            print(g, group_counters(data, prefixes, GROUP_UNITS[g]))
