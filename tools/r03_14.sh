#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=120
mkdir -p gpurun_out
Q="--steps 20 --warmup 3 --small-batches= --no-cpu-baseline --no-verify --no-device-half"
timeout 200 python bench.py $Q --hbm-gb 250 > gpurun_out/r03p_b250.log 2>&1
timeout 200 python bench.py $Q --hbm-gb 230 --depth 10 > gpurun_out/r03p_b230_d10.log 2>&1
