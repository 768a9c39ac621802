#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python bench.py --deliver --steps 12 --warmup 2 --no-cpu-baseline --no-verify --no-device-half --small-batches= > gpurun_out/r03x_deliver.log 2>&1
echo "rc=$?" >> gpurun_out/r03x_deliver.log
tail -c 200 gpurun_out/r03x_deliver.log
