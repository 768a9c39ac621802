#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=120
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/r03i_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03i_tests.log
tail -5 gpurun_out/r03i_tests.log
Q="--steps 20 --warmup 3 --small-batches= --no-cpu-baseline --no-verify --no-device-half"
timeout 400 python bench.py $Q > gpurun_out/r03i_b150.log 2>&1
timeout 400 python bench.py $Q --hbm-gb 200 > gpurun_out/r03i_b200.log 2>&1
timeout 400 python bench.py $Q --hbm-gb 250 > gpurun_out/r03i_b250.log 2>&1
