#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=60
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/r03c_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03c_tests.log
tail -8 gpurun_out/r03c_tests.log
Q="--small-batches= --no-cpu-baseline --no-verify --no-device-half --trace-memory"
timeout 500 python bench.py --steps 8 --warmup 2 $Q > gpurun_out/r03c_bench_150.log 2>&1; echo "rc=$?" >> gpurun_out/r03c_bench_150.log
timeout 500 python bench.py --steps 8 --warmup 2 --hbm-gb 250 $Q > gpurun_out/r03c_bench_250.log 2>&1; echo "rc=$?" >> gpurun_out/r03c_bench_250.log
timeout 500 python bench.py --steps 8 --warmup 2 --hbm-gb 250 --key-ahead 8 --depth 4 $Q > gpurun_out/r03c_bench_250_k8d4.log 2>&1; echo "rc=$?" >> gpurun_out/r03c_bench_250_k8d4.log
