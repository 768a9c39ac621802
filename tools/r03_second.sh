#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=60
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/r03b_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03b_tests.log
tail -8 gpurun_out/r03b_tests.log
timeout 600 python bench.py --steps 12 --warmup 2 --small-batches= --no-cpu-baseline > gpurun_out/r03b_bench.log 2>&1
echo "rc=$?" >> gpurun_out/r03b_bench.log
tail -c 1500 gpurun_out/r03b_bench.log
for W in 6 5 3; do
  ALFALFA_AMD_WGS_PER_CU=$W timeout 500 python bench.py --steps 12 --warmup 2 --small-batches= --no-cpu-baseline --no-verify --no-device-half > gpurun_out/r03b_bench_w$W.log 2>&1
  echo "rc=$?" >> gpurun_out/r03b_bench_w$W.log
done
