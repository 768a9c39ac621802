#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=120
mkdir -p gpurun_out
( echo "== 2200 streams x 12, default shape"; ALFALFA_AMD_TOKEN_PROFILE=1 timeout 300 python tools/parse_probe.py --streams 2200 --reps 1
  echo "== 480 streams x 12, default shape"; ALFALFA_AMD_TOKEN_PROFILE=1 timeout 300 python tools/parse_probe.py --streams 480 --reps 2
) > gpurun_out/r03h_probe.log 2>&1
Q="--steps 10 --warmup 2 --small-batches= --no-cpu-baseline --no-verify --no-device-half"
timeout 400 python bench.py $Q --hbm-gb 250 > gpurun_out/r03h_b250.log 2>&1
timeout 400 python bench.py $Q > gpurun_out/r03h_b150.log 2>&1
ALFALFA_AMD_WGS_PER_CU=6 timeout 400 python bench.py $Q --hbm-gb 250 > gpurun_out/r03h_b250_w6.log 2>&1
