#!/bin/bash
# Round 4, call w: the tests the stream re-keying touches (stochastic BSDF scenes under VCM / BDPT, stochastic alpha, heterogeneous media), the new
# reproducibility / pixel-shard / cross-process checkpoint tests, then configs[3] on two pixel-sharded contexts (with a second unsharded run as the yardstick).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4w
mkdir -p $O
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_pixel_sharding.py tests/test_gpu_checkpoint.py tests/test_gpu_parity.py tests/test_gpu_parity_hi.py tests/test_gpu_bdpt.py -q -m gpu \
  -k "pixel or checkpoint or stochastic_alpha or feature_scenes or all_bsdf or heterogeneous or (vcm_matches and (rough or glass or cloud)) or (bdpt_full and (glass or cloud))" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/log.txt
timeout 300 python tools/pixel_shard_study.py 8 > $O/round4_pixel_shard_configs3.json 2>> $O/err.txt
grep -v "^  File\|^Extension" $O/tests.log | tail -40; cat $O/log.txt; python3 -c "
import json; d=json.load(open('$O/round4_pixel_shard_configs3.json'))
for k in ('camera','light','camera, second unsharded run','light, second unsharded run'): print(k, d[k])
print([(c['context'], c['working_set_gb'], c['msamples_per_s']) for c in d['contexts']], d['rays_sum_equals_unsharded'])"
