#!/bin/bash
# round 3, first GPU contact of the queue-fed token workers: device-parse tests, then everything, then a short bench
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=40
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_device_parse.py -m gpu -x -q --timeout 200 > gpurun_out/r03a_t1.log 2>&1
echo "rc=$?" >> gpurun_out/r03a_t1.log
tail -5 gpurun_out/r03a_t1.log
if grep -q "rc=0" gpurun_out/r03a_t1.log; then
  timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r03a_tests.log 2>&1
  echo "rc=$?" >> gpurun_out/r03a_tests.log
  tail -15 gpurun_out/r03a_tests.log
fi
timeout 600 python bench.py --steps 10 --warmup 2 --small-batches= > gpurun_out/r03a_bench.log 2>&1
echo "rc=$?" >> gpurun_out/r03a_bench.log
tail -c 3000 gpurun_out/r03a_bench.log
