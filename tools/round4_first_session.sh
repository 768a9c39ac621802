#!/bin/bash
# First GPU session of the next round: what the end of round 3 left unmeasured, cheapest and most informative first.
# Each step writes to gpurun_out/ (copy what should be judged into profiles/).  Run the steps as separate gpurun calls if the
# budget is tight: every one is self-contained.
#
#   gpurun --timeout 900 -- 'bash tools/round4_first_session.sh 1 2'
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2 3 4 5} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }

# 1. one lane per DCT partition: never run on a GPU.  Parity first, then the latency of a 4-partition 1080p key frame
#    (<= 0.35 x the single-lane figure)
run 1 bash -c 'timeout 300 python tools/check_lane_per_partition.py --latency > gpurun_out/r04_lane_per_partition.log 2>&1; tail -4 gpurun_out/r04_lane_per_partition.log'

# 2. packed coefficient storage + the pool'"'"'s size classes together (each was run alone): the default bench command, packed
run 2 bash -c 'timeout 400 python bench.py --packed --steps 20 --warmup 5 > gpurun_out/r04_bench_packed.log 2>&1; tail -c 400 gpurun_out/r04_bench_packed.log'

# 3. the whole GPU suite with packed storage as the default of every context (tests that count dense chunks will need a look)
run 3 bash -c 'ALFALFA_AMD_PACKED=1 timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_gpu_tests_packed.log 2>&1; tail -5 gpurun_out/r04_gpu_tests_packed.log'

# 4. ... and with a lane per partition allowed everywhere
run 4 bash -c 'ALFALFA_AMD_LANE_PER_PARTITION=1 timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_gpu_tests_lpp.log 2>&1; tail -5 gpurun_out/r04_gpu_tests_lpp.log'

# 5. the four-partition benchmark config end to end, one lane per frame vs one lane per partition
run 5 bash -c 'timeout 300 python bench.py --config 1080p_inter_lf_subpel --steps 8 --warmup 2 --no-cpu-baseline --small-batches "" > gpurun_out/r04_bench_subpel.log 2>&1; ALFALFA_AMD_LANE_PER_PARTITION=1 timeout 300 python bench.py --config 1080p_inter_lf_subpel --steps 8 --warmup 2 --no-cpu-baseline --small-batches "" > gpurun_out/r04_bench_subpel_lpp.log 2>&1; tail -c 300 gpurun_out/r04_bench_subpel.log; tail -c 300 gpurun_out/r04_bench_subpel_lpp.log'
