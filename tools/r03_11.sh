#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=120
mkdir -p gpurun_out
Q="--steps 20 --warmup 3 --small-batches= --no-cpu-baseline --no-verify --no-device-half"
timeout 400 python bench.py $Q --hbm-gb 200 > gpurun_out/r03m_b200.log 2>&1
timeout 400 python bench.py $Q --hbm-gb 200 --overcommit 1.5 > gpurun_out/r03m_b200_oc15.log 2>&1
timeout 400 python bench.py $Q --hbm-gb 150 --key-ahead 10 --overcommit 1.4 > gpurun_out/r03m_b150_k10.log 2>&1
timeout 300 python -m pytest tests/test_gpu_device_parse.py -m gpu -q --timeout 300 -x -k "heap_runs_out or context_info" > gpurun_out/r03m_tests.log 2>&1
tail -3 gpurun_out/r03m_tests.log
