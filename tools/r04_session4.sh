#!/bin/bash
# Round 4, fourth GPU session: what the box's host really gives (CPU quota?), the default bench command after the pool fixes
# (compute-ordered recycling of rasters / dense transients, adaptive host share), the whole GPU suite.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2 3} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
run 1 bash -c 'timeout 200 python tools/host_parallelism.py > gpurun_out/r04d_host_parallelism.log 2>&1; cat gpurun_out/r04d_host_parallelism.log'
run 2 bash -c 'timeout 420 python bench.py --steps 20 --warmup 5 --trace-memory > gpurun_out/r04d_bench.log 2> gpurun_out/r04d_bench.err; echo rc=$?; grep "^\[bench" gpurun_out/r04d_bench.err | cut -c1-400; tail -2 gpurun_out/r04d_bench.err | cut -c1-300'
run 3 bash -c 'timeout 1200 python -m pytest tests -q -m gpu --timeout 600 > gpurun_out/r04d_gpu_tests.log 2>&1; echo rc=$?; tail -8 gpurun_out/r04d_gpu_tests.log | cut -c1-300'
run 4 bash -c 'timeout 300 python bench.py --steps 12 --warmup 3 --deliver --secondary "" --small-batches "" --no-cpu-baseline --lanes-only-steps 0 > gpurun_out/r04d_bench_deliver.log 2> gpurun_out/r04d_bench_deliver.err; echo rc=$?; grep "^\[bench" gpurun_out/r04d_bench_deliver.err | cut -c1-300'
