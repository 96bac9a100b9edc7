#!/bin/bash
# Instruction cost of the parts of k_camera_shade by duplication (dev_vcm_steps.h, ETX_HIP_COST_PROBE): one PMC pass per
# debug flag over a one-lane bench run of the probe variant; prints per-kernel instruction sums.
set -u
root=$(pwd)
out=$root/gpurun_out/cost_probe
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$root/etx-tracer_amd/variants/libetx_hip_probe.so
for flags in 0 0x400 0x800 0x1000 0x2000 0x4000 0x100 0x200 0x300; do
  ETX_HIP_DEBUG_FLAGS=$flags rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/f$flags -o pmc -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-table > $out/f$flags.log 2>&1
  python3 $root/tools/pmc_aggregate.py $out/f$flags.json $(find $out/f$flags -name "*counter_collection.csv") > /dev/null
  python3 - <<PY
import json
d = json.load(open("$out/f$flags.json"))
k = d["k_camera_shade<0u, false>"]
print("flags %-7s camera_shade: dur %8.0f us  VALU %.4g  SALU %.4g  VMEM %.4g  LDS %.4g" % ("$flags", k["duration_us_sum"], k["SQ_INSTS_VALU_sum"], k["SQ_INSTS_SALU_sum"], k["SQ_INSTS_VMEM_sum"], k["SQ_INSTS_LDS_sum"]))
PY
  rm -rf $out/f$flags
done
