#!/bin/bash
# Round 4, second GPU session: the row-per-token probe, the default bench command with progress on stderr (the first session's run
# was cut off at 600 s without a line), the tests that failed / did not exist in session 1.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2 3} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
run 1 bash -c 'PROBE_ROWS_ONLY=1 timeout 120 gpurun_in/probe_probs > gpurun_out/r04b_probe_rows.log 2>&1; grep "waves 1024" gpurun_out/r04b_probe_rows.log'
run 2 bash -c 'timeout 420 python bench.py --steps 20 --warmup 5 --trace-memory > gpurun_out/r04b_bench.log 2> gpurun_out/r04b_bench.err; echo rc=$?; grep "^\[bench" gpurun_out/r04b_bench.err | cut -c1-400; tail -3 gpurun_out/r04b_bench.err | cut -c1-300; tail -c 600 gpurun_out/r04b_bench.log'
run 3 bash -c 'timeout 900 python -m pytest tests/test_gpu_lane_per_partition.py tests/test_gpu_packed_coefficients.py tests/test_gpu_device_parse.py tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "lane_per_partition or packed or heap or key_frames or info or delivered or small_calls" -rP > gpurun_out/r04b_gpu_tests.log 2>&1; echo rc=$?; grep -n "1080p key frame\|passed\|failed" gpurun_out/r04b_gpu_tests.log | tail -5; grep -n "^FAILED\|^ERROR" gpurun_out/r04b_gpu_tests.log | head -20'
run 4 bash -c 'timeout 300 python bench.py --steps 12 --warmup 3 --host-share-ms 0 --hbm-gb 200 --secondary "" --small-batches "" --no-cpu-baseline --lanes-only-steps 0 > gpurun_out/r04b_bench_lanes200.log 2> gpurun_out/r04b_bench_lanes200.err; echo rc=$?; grep "^\[bench" gpurun_out/r04b_bench_lanes200.err | cut -c1-300'
