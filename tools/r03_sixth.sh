#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=60
mkdir -p gpurun_out
Q="--steps 8 --warmup 2 --small-batches= --no-cpu-baseline --no-verify --no-device-half"
timeout 400 python bench.py $Q --hbm-gb 250 > gpurun_out/r03f_b250.log 2>&1
ALFALFA_AMD_WGS_PER_CU=6 timeout 400 python bench.py $Q --hbm-gb 250 > gpurun_out/r03f_b250_w6.log 2>&1
timeout 400 python bench.py $Q > gpurun_out/r03f_b150.log 2>&1
