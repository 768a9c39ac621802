#!/bin/bash
# Round 4, fifth GPU session: default bench (urgent key frames on the host route, worker threads bounded by the CPU quota), the tests
# that failed in session 4 (pieces released while binding) + the new xc-dump caller, then the rocprofv3 passes.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2 3} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
run 1 bash -c 'timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/r04e_bench.log 2> gpurun_out/r04e_bench.err; echo rc=$?; grep "^\[bench" gpurun_out/r04e_bench.err | cut -c1-400; tail -2 gpurun_out/r04e_bench.err | cut -c1-300'
run 2 bash -c 'timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_parse.py tests/test_gpu_lf_search.py tests/test_reference_callers.py -q -m gpu --timeout 600 > gpurun_out/r04e_gpu_tests.log 2>&1; echo rc=$?; tail -8 gpurun_out/r04e_gpu_tests.log | cut -c1-300'
run 3 bash tools/r04_profile.sh 1 2 3
