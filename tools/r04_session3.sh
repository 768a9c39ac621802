#!/bin/bash
# Round 4, third GPU session: the default bench command (host arenas of one size, admission that knows pending pieces), the whole
# GPU suite.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
run 1 bash -c 'timeout 420 python bench.py --steps 20 --warmup 5 --trace-memory > gpurun_out/r04c_bench.log 2> gpurun_out/r04c_bench.err; echo rc=$?; grep "^\[bench" gpurun_out/r04c_bench.err | cut -c1-400; tail -3 gpurun_out/r04c_bench.err | cut -c1-300'
run 2 bash -c 'timeout 1200 python -m pytest tests -q -m gpu --timeout 600 > gpurun_out/r04c_gpu_tests.log 2>&1; echo rc=$?; tail -8 gpurun_out/r04c_gpu_tests.log | cut -c1-300'
run 3 bash -c 'timeout 300 python bench.py --steps 12 --warmup 3 --deliver --secondary "" --small-batches "" --no-cpu-baseline --lanes-only-steps 0 > gpurun_out/r04c_bench_deliver.log 2> gpurun_out/r04c_bench_deliver.err; echo rc=$?; grep "^\[bench" gpurun_out/r04c_bench_deliver.err | cut -c1-300'
