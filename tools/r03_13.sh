#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=120
mkdir -p gpurun_out
Q="--steps 20 --warmup 3 --small-batches= --no-cpu-baseline --no-verify --no-device-half"
timeout 240 python bench.py $Q > gpurun_out/r03o_b200.log 2>&1
