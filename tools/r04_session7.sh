#!/bin/bash
# Round 4, seventh GPU session: the driver's commands on the final tree -- default bench, whole GPU suite.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
run 1 bash -c 'timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/r04g_bench.log 2> gpurun_out/r04g_bench.err; echo rc=$?; grep "^\[bench\|Error" gpurun_out/r04g_bench.err | cut -c1-330'
run 2 bash -c 'timeout 1200 python -m pytest tests -q -m gpu --timeout 600 > gpurun_out/r04g_gpu_tests.log 2>&1; echo rc=$?; tail -8 gpurun_out/r04g_gpu_tests.log | cut -c1-300'
run 3 bash -c 'timeout 300 python bench.py --steps 20 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 > gpurun_out/r04g_bench_again.log 2> gpurun_out/r04g_bench_again.err; echo rc=$?; grep "timed region\|Error" gpurun_out/r04g_bench_again.err | cut -c1-330'
