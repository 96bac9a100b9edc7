"""CPU: the counter summaries bench.py quotes belong to the shipped build.

bench.py fills `kernels[*].counters_1lane` and `roofline.traffic` from profiles/round*_pmc_<workload>_1lane_summary.json (tools/profile_lookup.py:
the newest summary whose `_meta` names the workload) - but only when the summary's `_meta.library_sha16` is the hash of the library it has
loaded; otherwise it prints `"counters_stale": true` and withholds the counters (round 5; until round 4 only kernel NAMES were compared).
Here: a summary that claims the shipped library's hash must name only kernels that library holds, and every summary carries the meta data the
lookup divides by. Which workloads have a summary of the shipped build is printed (a stale one is legal - the bench line says so - but
tools/profile_round.sh should be run again before a round closes)."""
import hashlib
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_summaries_of_the_shipped_library_name_its_kernels():
    from tools import profile_lookup
    from etx_tracer_amd import api
    library = profile_lookup.library_kernels()
    assert "k_vcm_commit" in library and any(name.startswith("k_trace_closest<") for name in library)
    with open(api.library_path(), "rb") as f:
        shipped = hashlib.sha256(f.read()).hexdigest()[:16]
    newest = {}
    for rnd, path, data in profile_lookup.summaries():
        newest.setdefault(data["_meta"].get("workload"), (path, data))
    for workload, (path, data) in newest.items():
        meta = data["_meta"]
        assert meta.get("iterations", 0) >= 1 and meta.get("units_per_step", {}).get("rays_extension", 0) > 0, path
        current = meta.get("library_sha16") == shipped
        print("%-16s %s: %s" % (workload, os.path.relpath(path, ROOT), "of the shipped library" if current else "STALE (library %s, shipped %s): bench.py withholds its counters" % (meta.get("library_sha16"), shipped)))
        if current:
            ours = [n for n in data if (n != "_meta") and n.startswith("k_")]  # vendor kernels (hipcub sort of the device tree build, memset) are not ours to find
            missing = [n for n in ours if n not in library]
            assert missing == [], "%s names kernels the shipped library does not hold: %s" % (os.path.relpath(path, ROOT), missing)


def test_lookup_reports_missing_as_none():
    from tools import profile_lookup
    assert profile_lookup.group_counters(None, ("k_trace_closest",), "rays_extension") is None
    summary = {"_meta": {"workload": "x", "iterations": 2, "units_per_step": {"rays_extension": 100.0}},
               "k_trace_closest<true, true, true>": {"launches": 4, "duration_us_sum": 10.0, "SQ_INSTS_VALU_sum": 300.0, "FETCH_SIZE_sum": 1.0, "WRITE_SIZE_sum": 2.0}}
    row = profile_lookup.group_counters(summary, ("k_trace_closest",), "rays_extension")
    assert row["kernel"] == "k_trace_closest<true, true, true>" and row["valu_lane_instructions_per_unit"] == 96.0
    assert row["hbm_bytes_per_step"] == 2048 and row["hbm_bytes_per_unit"] == 20.48 and row["l2_hit_rate"] is None and row["waves_waiting_share"] is None
    assert profile_lookup.group_counters(summary, ("k_merge_",), "photons_examined") is None
