#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=60
mkdir -p gpurun_out
Q="--steps 6 --warmup 1 --hbm-gb 250 --key-ahead 8 --depth 4 --small-batches= --no-cpu-baseline --no-verify --no-device-half"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $Q > gpurun_out/r03d_$name.log 2>&1; echo "rc=$?" >> gpurun_out/r03d_$name.log; }
run novmm ALFALFA_AMD_NO_VMM=1 ALFALFA_AMD_HEAP_LIMIT_MB=131072
run vmm_align1g ALFALFA_AMD_HEAP_VA_ALIGN_MB=1024
run vmm_w6 ALFALFA_AMD_WGS_PER_CU=6
run novmm_w6 ALFALFA_AMD_NO_VMM=1 ALFALFA_AMD_HEAP_LIMIT_MB=131072 ALFALFA_AMD_WGS_PER_CU=6
