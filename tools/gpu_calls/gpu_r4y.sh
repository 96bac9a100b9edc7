#!/bin/bash
# Round 4, call y: the shade step cut in two - the simple group's emitter / camera connections as endpoint requests evaluated by k_connect_endpoints<.., true>
# (-DETX_SIMPLE_NEE_QUEUE=1, etx-tracer_amd/variants/libetx_hip_neeq.so) against the product build; parity of the variant first.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4y_${ETX_AB_VARIANT:-neeq}
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
ETX_HIP_LIBRARY=$V/libetx_hip_${ETX_AB_VARIANT:-neeq}.so timeout 200 python -m pytest tests/test_gpu_parity_hi.py tests/test_gpu_parity.py -q -m gpu -k "shared_streams or vcm_full_cornell or vcm_classic_cornell" 2>&1 | tail -4 > $O/parity.txt
for round in 1 2; do
  for v in product ${ETX_AB_VARIANT:-neeq}; do
    lib=$V/libetx_hip_$v.so; [ $v = product ] && lib=etx-tracer_amd/libetx_hip.so
    r=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    c=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload classic --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    r1=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "$v: full $r, classic $c, full on one lane $r1" >> $O/ab.txt
  done
done
cat $O/parity.txt; cat $O/ab.txt; tail -2 $O/err.txt
