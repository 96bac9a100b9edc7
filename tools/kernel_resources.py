#!/usr/bin/env python3
"""Per-kernel and per-function register budget of the gfx950 code (VGPRs, AGPRs, spills, scratch, code size), read from
the assembly hipcc emits for one translation unit of etx-tracer_amd/csrc:

    python3 tools/kernel_resources.py kernels_shade_camera_general.hip [more.hip ...] [--json out.json]

Used by tests/test_build_budget.py (register budget of the bench-path and general-material kernels)."""
import json
import os
import re
import subprocess
import sys
import tempfile
import concurrent.futures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "etx-tracer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-x", "hip", "--offload-device-only", "-S"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def analyse(source):
    path = source if os.path.isabs(source) else os.path.join(CSRC, source)
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as tmp:
        asm = tmp.name
    extra = os.environ.get("ETX_HIP_EXTRA_FLAGS", "").split()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [path, "-o", asm], stderr=subprocess.DEVNULL)
    text = open(asm).read()
    os.remove(asm)
    rows = []
    current = None
    for line in text.split("\n"):
        m = re.match(r"\s*\.type\s+(\S+),@function", line)
        if m:
            current = {"symbol": m.group(1), "kernel": False}
            rows.append(current)
            continue
        if current is None:
            continue
        for key, pattern in (("code_bytes", r"; codeLenInByte = (\d+)"), ("vgprs", r"; NumVgprs: (\d+)"), ("agprs", r"; NumAgprs: (\d+)"), ("total_vgprs", r"; TotalNumVgprs: (\d+)"),
                             ("scratch", r"; ScratchSize: (\d+)"), ("sgprs", r"; NumSgprs: (\d+)"), ("occupancy", r"; Occupancy: (\d+)")):
            mm = re.match(pattern, line)
            if mm:
                current[key] = int(mm.group(1))
    # kernel metadata (spill counts)
    for block in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, flags=re.S):
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        for row in rows:
            if row["symbol"] == name:
                row["kernel"] = True
                row["sgpr_spills"] = int(re.search(r"\.sgpr_spill_count:\s+(\d+)", block).group(1))
                row["vgpr_spills"] = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1))
                row["lds"] = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", block).group(1))
    names = demangle([r["symbol"] for r in rows])
    for r in rows:
        r["name"] = re.sub(r"\(.*", "", names.get(r["symbol"], r["symbol"]))
        r["source"] = os.path.basename(source)
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    json_out = None
    if "--json" in sys.argv:
        json_out = sys.argv[sys.argv.index("--json") + 1]
        args.remove(json_out)
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        results = list(pool.map(analyse, args))
    rows = [r for rs in results for r in rs]
    print("%-28s %-62s %5s %5s %6s %7s %6s %6s %8s" % ("source", "function", "vgpr", "agpr", "total", "scratch", "vspill", "sspill", "code"))
    for r in rows:
        print("%-28s %-62s %5d %5d %6d %7d %6s %6s %8d" % (r["source"][:28], ("K " if r["kernel"] else "  ") + r["name"][-60:], r.get("vgprs", 0), r.get("agprs", 0), r.get("total_vgprs", 0),
                                                       r.get("scratch", 0), r.get("vgpr_spills", "-"), r.get("sgpr_spills", "-"), r.get("code_bytes", 0)))
    if json_out:
        with open(json_out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
