#!/bin/bash
# round 6, GPU call o: (1) iterations in flight for the bidirectional integrator now that its kernels are narrower: ETX_HIP_LANES 6 (default) vs 8 on configs[3] / configs[4];
# (2) occlusion traversal that enters the nearest child first (experiment build of kernels_trace.hip, -DETX_OCCLUDED_NEAR_FIRST) vs slot order, configs[3] and the configs[2] family.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r6o
mkdir -p $O
export TMPDIR=/tmp
NEAR=$PWD/etx-tracer_amd/variants/libetx_hip_nearfirst.so
BASE=$PWD/etx-tracer_amd/libetx_hip.so
ETX_HIP_LIBRARY=$NEAR timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sssmesh.py -x -q -m gpu -p no:cacheprovider -k "trace or ray or shadow or transmittance or gems or bidirectional or bdpt" > $O/tests_near.log 2>&1
echo "near-first library: ray queries + tree films rc=$? $(grep -E 'passed|failed|error' $O/tests_near.log | tail -1)" >> $O/log.txt
run() { # label workload steps env...
  label=$1; w=$2; steps=$3; shift 3
  x=$(env "$@" timeout 400 python3 bench.py --workload $w --steps $steps --warmup 4 --repeats 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['repeats']['values'], 'lanes', d['config'].get('lanes'), 'GB', d['config'].get('working_set_gb'))")
  echo "$w $label: $x" >> $O/ab.txt
}
for r in 1 2; do
  run "base run $r" sssdragon_bdpt 8 ETX_HIP_LIBRARY=$BASE
  run "nearfirst run $r" sssdragon_bdpt 8 ETX_HIP_LIBRARY=$NEAR
  run "base 8 lanes run $r" sssdragon_bdpt 8 ETX_HIP_LIBRARY=$BASE ETX_HIP_LANES=8
  run "base run $r" gems 12 ETX_HIP_LIBRARY=$BASE
  run "nearfirst run $r" gems 12 ETX_HIP_LIBRARY=$NEAR
  run "base run $r" cloud_bdpt 8 ETX_HIP_LIBRARY=$BASE
  run "base 8 lanes run $r" cloud_bdpt 8 ETX_HIP_LIBRARY=$BASE ETX_HIP_LANES=8
done
cat $O/log.txt $O/ab.txt
