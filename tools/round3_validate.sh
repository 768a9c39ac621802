#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r03v_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03v_gpu_tests.log
tail -12 gpurun_out/r03v_gpu_tests.log
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r03v_bench.log 2>&1
echo "rc=$?" >> gpurun_out/r03v_bench.log
tail -c 300 gpurun_out/r03v_bench.log
