#!/bin/bash
# Round 4, first GPU session: the probe behind "64 chains per wave", the default bench command (packed storage, 150 GB), the
# whole GPU suite (both coefficient formats, lane per partition, two ranks on one GPU).  Every step under its own timeout.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2 3} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
run 1 bash -c 'timeout 120 gpurun_in/probe_probs > gpurun_out/r04a_probe_probs.log 2>&1; tail -30 gpurun_out/r04a_probe_probs.log'
run 2 bash -c 'timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.log 2> gpurun_out/r04a_bench.err; echo rc=$?; tail -c 1500 gpurun_out/r04a_bench.log; tail -5 gpurun_out/r04a_bench.err'
run 3 bash -c 'timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/r04a_gpu_tests.log 2>&1; echo rc=$?; tail -25 gpurun_out/r04a_gpu_tests.log'
