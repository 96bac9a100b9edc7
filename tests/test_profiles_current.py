"""CPU: the counter summaries bench.py quotes belong to the shipped build.

bench.py fills `kernels[*].counters_1lane` and `roofline.traffic` from profiles/round*_pmc_<workload>_1lane_summary.json (tools/profile_lookup.py:
the newest summary whose `_meta` names the workload). Every kernel such a summary names must exist in etx-tracer_amd/libetx_hip.so - a summary
of an earlier build whose kernels have been renamed or re-templated is stale evidence and fails here instead of being printed."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_summaries_name_kernels_of_the_shipped_library():
    from tools import profile_lookup
    library = profile_lookup.library_kernels()
    assert "k_vcm_commit" in library and any(name.startswith("k_trace_closest<") for name in library)
    newest = {}
    for rnd, path, data in profile_lookup.summaries():
        newest.setdefault(data["_meta"].get("workload"), (path, data))
    for workload, (path, data) in newest.items():
        names = [n for n in data if n != "_meta"]
        ours = [n for n in names if n.startswith("k_")]  # vendor kernels (hipcub sort of the device tree build, memset) are not ours to find
        missing = [n for n in ours if n not in library]
        assert missing == [], "%s names kernels the shipped library does not hold: %s" % (os.path.relpath(path, ROOT), missing)
        meta = data["_meta"]
        assert meta.get("iterations", 0) >= 1 and meta.get("units_per_step", {}).get("rays_extension", 0) > 0, path


def test_lookup_reports_missing_as_none():
    from tools import profile_lookup
    assert profile_lookup.group_counters(None, ("k_trace_closest",), "rays_extension") is None
    summary = {"_meta": {"workload": "x", "iterations": 2, "units_per_step": {"rays_extension": 100.0}},
               "k_trace_closest<true, true, true>": {"launches": 4, "duration_us_sum": 10.0, "SQ_INSTS_VALU_sum": 300.0, "FETCH_SIZE_sum": 1.0, "WRITE_SIZE_sum": 2.0}}
    row = profile_lookup.group_counters(summary, ("k_trace_closest",), "rays_extension")
    assert row["kernel"] == "k_trace_closest<true, true, true>" and row["valu_lane_instructions_per_unit"] == 96.0
    assert row["hbm_bytes_per_step"] == 2048 and row["hbm_bytes_per_unit"] == 20.48 and row["l2_hit_rate"] is None and row["waves_waiting_share"] is None
    assert profile_lookup.group_counters(summary, ("k_merge_",), "photons_examined") is None
