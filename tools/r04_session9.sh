#!/bin/bash
# Round 4, ninth GPU session: the final tree's default bench line + the routing tests.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04i_bench.log 2> gpurun_out/r04i_bench.err; echo rc=$?; grep "^\[bench\|Error" gpurun_out/r04i_bench.err | cut -c1-300
timeout 150 python -m pytest tests/test_gpu_device_parse.py -q -m gpu --timeout 120 -k "key_frames or small_calls or info" > gpurun_out/r04i_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r04i_tests.log
