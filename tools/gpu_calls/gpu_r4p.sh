#!/bin/bash
# Round 4, call p: A/B of the medium rows + the scene-level normal-map / texture switches + the light vertex read from its record in plain Lambert
# scenes (base = the build before them); the path table length (k_expand_pairs walks the list beyond it); the gather microbenchmark's
# few-records-per-wave rows; the pixel-shard study with relative differences.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4p
mkdir -p $O
export TMPDIR=/tmp
V=etx-tracer_amd/variants
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
timeout 60 tools/micro/bin/gather_bench > $O/gather_bench.txt 2>&1
for round in 1 2; do
  for v in base new; do
    lib=$V/libetx_hip_$v.so; [ $v = new ] && lib=etx-tracer_amd/libetx_hip.so
    for w in full; do
      r=$(ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload $w --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
      echo "$w $v 4 lanes: $r" >> $O/ab.txt
    done
    r=$(ETX_HIP_LANES=1 ETX_HIP_LIBRARY=$lib timeout 120 python bench.py --workload full --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full $v 1 lane: $r" >> $O/ab.txt
  done
done
for round in 1; do
  for t in 8 16 32; do
    r=$(ETX_HIP_PATH_TABLE=$t ETX_HIP_LIBRARY=$V/libetx_hip_dbgapi.so timeout 120 python bench.py --workload full --steps 24 --warmup 8 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    r1=$(ETX_HIP_LANES=1 ETX_HIP_PATH_TABLE=$t ETX_HIP_LIBRARY=$V/libetx_hip_dbgapi.so timeout 120 python bench.py --workload full --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-table 2>>$O/err.txt | line)
    echo "full path table $t: 4 lanes $r, 1 lane $r1" >> $O/ab.txt
  done
done
timeout 300 python tools/pixel_shard_study.py 8 > $O/round4_pixel_shard_configs3.json 2>> $O/err.txt
cat $O/ab.txt; tail -3 $O/err.txt; grep "records per wave\|no load\|ds_read_b32\|ds_read_b64" $O/gather_bench.txt; grep -A8 '"camera"' $O/round4_pixel_shard_configs3.json
