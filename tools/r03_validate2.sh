#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-verify --no-device-half > gpurun_out/r03w_bench.log 2>&1
echo "rc=$?" >> gpurun_out/r03w_bench.log
timeout 100 python -m pytest tests/test_bench_smoke.py -m gpu -q --timeout 100 > gpurun_out/r03w_smoke.log 2>&1
tail -3 gpurun_out/r03w_smoke.log
