#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=120
mkdir -p gpurun_out
Q="--steps 20 --warmup 3 --small-batches= --no-cpu-baseline --no-verify --no-device-half --trace-memory"
timeout 400 python bench.py $Q --hbm-gb 200 > gpurun_out/r03l_b200.log 2>&1
timeout 400 python bench.py $Q > gpurun_out/r03l_b150.log 2>&1
timeout 400 python bench.py $Q --hbm-gb 200 --overcommit 1.6 > gpurun_out/r03l_b200_oc16.log 2>&1
