#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection.csv files per kernel.

    python3 tools/pmc_aggregate.py out.json pass1_counter_collection.csv [pass2_... ...]

Every pass holds a few counters (the SQ / TCC slots of gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"); the passes ran
the same command, so per-kernel sums are comparable between passes. Output: {kernel: {launches, duration_us_sum,
<counter>_sum, ...}} with kernel names shortened to the template name. Derived figures are added where their inputs are
present: l2_hit_rate, wait_share / issue_stall_share / active_share (of SQ_WAVE_CYCLES), valu_share.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short_name(name):
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:  # cut the argument list, keep the template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).replace("etxd::", "")


def main():
    # --meta <file>: a log of the profiled command whose last JSON line is bench.py's; its workload, step counts and units of work per step go
    # into the summary (`_meta`) so that readers (tools/profile_lookup.py, bench.py) divide these counters by THIS run's units, not another's
    meta = None
    if "--meta" in sys.argv:
        i = sys.argv.index("--meta")
        meta_path = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        line = None
        with open(meta_path) as f:
            for text in f:
                text = text.strip()
                if text.startswith("{") and '"metric"' in text:
                    line = json.loads(text)
        if line is not None:
            meta = {"workload": line.get("config", {}).get("workload_key"), "iterations": int(line["steps"]) + int(line["warmup"]), "lanes": line.get("config", {}).get("lanes"),
                    "units_per_step": line.get("counters", {}).get("units_per_step", {}), "ms_per_step": line.get("ms_per_step"), "value": line.get("value"),
                    "library_sha16": line.get("config", {}).get("library_sha16")}
    verbose = "--all" in sys.argv  # print every counter sum, not only the derived shares
    if verbose:
        sys.argv.remove("--all")
    out_path, paths = sys.argv[1], sys.argv[2:]
    table = defaultdict(lambda: defaultdict(float))
    passes = defaultdict(set)  # counter -> the passes that collected it (a counter listed in two passes is averaged, not added)
    for path in paths:
        seen = set()
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short_name(row["Kernel_Name"])
                table[k][row["Counter_Name"] + "_sum"] += float(row["Counter_Value"])
                passes[row["Counter_Name"] + "_sum"].add(path)
                table[k]["vgprs@"] = float(row.get("VGPR_Count", 0) or 0)
                table[k]["agprs@"] = float(row.get("Accum_VGPR_Count", 0) or 0)
                table[k]["scratch@"] = float(row.get("Scratch_Size", 0) or 0)
                key = (row["Dispatch_Id"], k)
                if key not in seen:
                    seen.add(key)
                    table[k]["launches@" + path] += 1
                    table[k]["duration_us@" + path] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1.0e-3
    result = {}
    for k, v in table.items():
        launches = max(val for key, val in v.items() if key.startswith("launches@"))
        duration = max(val for key, val in v.items() if key.startswith("duration_us@"))
        r = {"launches": int(launches), "duration_us_sum": round(duration, 1), "vgprs": int(v["vgprs@"]), "agprs": int(v["agprs@"]), "scratch_bytes": int(v["scratch@"])}
        for key, val in sorted(v.items()):
            if "@" not in key:
                r[key] = round(val / max(1, len(passes[key])), 3)
        if "TCC_HIT_sum_sum" in r and (r["TCC_HIT_sum_sum"] + r.get("TCC_MISS_sum_sum", 0.0)) > 0:
            r["l2_hit_rate"] = round(r["TCC_HIT_sum_sum"] / (r["TCC_HIT_sum_sum"] + r["TCC_MISS_sum_sum"]), 4)
        wc = r.get("SQ_WAVE_CYCLES_sum", 0.0)
        if wc > 0:
            for src, dst in (("SQ_WAIT_ANY_sum", "wait_share"), ("SQ_WAIT_INST_ANY_sum", "issue_stall_share"), ("SQ_ACTIVE_INST_ANY_sum", "active_share"),
                             ("SQ_ACTIVE_INST_VALU_sum", "valu_share"), ("SQ_ACTIVE_INST_VMEM_sum", "vmem_share"), ("SQ_ACTIVE_INST_LDS_sum", "lds_share"),
                             ("SQ_ACTIVE_INST_SCA_sum", "scalar_share")):
                if src in r:
                    r[dst] = round(r[src] / wc, 4)
        if r.get("SQC_ICACHE_REQ_sum", 0) > 0:
            r["icache_miss_rate"] = round(r.get("SQC_ICACHE_MISSES_sum", 0.0) / r["SQC_ICACHE_REQ_sum"], 4)
        if r.get("SQC_DCACHE_REQ_sum", 0) > 0:
            r["scalar_dcache_miss_rate"] = round(r.get("SQC_DCACHE_MISSES_sum", 0.0) / r["SQC_DCACHE_REQ_sum"], 4)
        if r.get("TCP_TCC_READ_REQ_sum_sum", 0) > 0:
            r["l1_read_latency_mean"] = round(r.get("TCP_TCC_READ_REQ_LATENCY_sum_sum", 0.0) / r["TCP_TCC_READ_REQ_sum_sum"], 1)
        if r.get("TCP_TOTAL_CACHE_ACCESSES_sum_sum", 0) > 0 and "TCP_TCC_READ_REQ_sum_sum" in r:
            r["l1_to_l2_read_rate"] = round(r["TCP_TCC_READ_REQ_sum_sum"] / r["TCP_TOTAL_CACHE_ACCESSES_sum_sum"], 4)
        for name in ("OccupancyPercent", "VALUBusy", "MemUnitStalled", "MemUnitBusy", "LDSBankConflict"):
            if name + "_sum" in r:
                r[name + "_mean"] = round(r[name + "_sum"] / launches, 3)
        result[k] = r
    total = sum(r["duration_us_sum"] for r in result.values())
    for r in result.values():
        r["time_share"] = round(r["duration_us_sum"] / total, 4) if total else 0.0
    ordered = dict(sorted(result.items(), key=lambda kv: -kv[1]["duration_us_sum"]))
    if meta is not None:
        ordered = dict([("_meta", meta)] + list(ordered.items()))
    with open(out_path, "w") as f:
        json.dump(ordered, f, indent=1)
    for k, r in sorted(result.items(), key=lambda kv: -kv[1]["duration_us_sum"])[:14]:
        print("%-44s %5.1f%% " % (k[:44], 100.0 * r["time_share"]), {a: b for a, b in r.items() if a.endswith(("share", "rate", "mean")) or (verbose and a.endswith("_sum"))})


if __name__ == "__main__":
    main()
