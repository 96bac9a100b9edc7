#!/bin/bash
# round 4, call o: the pixel-sharding tests and the > 8-hit continuous_trace probe on the device (neither has run on a GPU yet);
# configs[3] on two pixel-sharded contexts of the one device
set -u
mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_gpu_pixel_sharding.py tests/test_continuous_trace_probe.py tests/test_gpu_checkpoint.py -q -m gpu 2>&1 | tail -40 | tee gpurun_out/r4o/tests.txt
timeout 400 python tools/pixel_shard_study.py 8 > gpurun_out/r4o/round4_pixel_shard_configs3.json 2> gpurun_out/r4o/study_err.txt
tail -5 gpurun_out/r4o/study_err.txt; cat gpurun_out/r4o/round4_pixel_shard_configs3.json
