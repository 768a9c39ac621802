#!/bin/bash
# Round 4 profile on the MI355X box.  The token workers are RESIDENT (they draw frames from a queue and linger between bursts), so a
# counter pass -- which serialises dispatches and needs every dispatch to END -- runs them in a counter-friendly mode:
# ALFALFA_AMD_WORKER_LINGER_MS=0 (a wave leaves as soon as it has no frame and the queue is empty) and the entropy decode ALONE
# (tools/parse_probe.py: 96 streams x 12 frames, every frame on the lanes).  Every counter set in its own pass (gpurun refuses
# --pmc together with the trace domains that crash nodes).  Writes gpurun_out/prof_r04/ (scratch); tools/r04_pmc_summary.py turns it
# into profiles/r04_*.md and profiles/pmc_traffic.json.      bash tools/r04_profile.sh [steps...]
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
want=" ${*:-1 2 3} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
# 1. kernel trace + stats of the default bench command, shortened (the dominant kernel's durations must agree with bench.py's own figures)
run 1 bash -c "timeout 400 rocprofv3 --kernel-trace --stats -d $O -o kt -- python $R/bench.py --steps 8 --warmup 2 --secondary '' --small-batches '' --no-cpu-baseline --lanes-only-steps 0 > $O/kt.log 2> $O/kt.err; echo rc=\$?; ls $O | head -20"
# 2. the entropy decode alone, counter-friendly: HBM traffic and issue counters of k_token_workers / k_parse_mb_headers
P="python $R/tools/parse_probe.py --streams 96 --frames 12 --reps 1"
run 2 bash -c "export ALFALFA_AMD_WORKER_LINGER_MS=0 ALFALFA_AMD_ROUTE=device; timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o tok_fetch -- $P > $O/tok_fetch.log 2>&1; timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o tok_write -- $P > $O/tok_write.log 2>&1; timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O -o tok_sq -- $P > $O/tok_sq.log 2>&1; tail -2 $O/tok_fetch.log $O/tok_write.log $O/tok_sq.log; ls $O"
# 3. the reconstruction kernels: a shallow pipeline of the bench (traffic per macroblock does not depend on its depth)
B="python $R/bench.py --steps 2 --warmup 0 --key-ahead 2 --depth 2 --secondary= --small-batches= --no-cpu-baseline --no-verify --no-device-half --lanes-only-steps 0 --streams 240"
run 3 bash -c "export ALFALFA_AMD_WORKER_LINGER_MS=0; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o rec_fetch -- $B > $O/rec_fetch.log 2>&1; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o rec_write -- $B > $O/rec_write.log 2>&1; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O -o rec_sq -- $B > $O/rec_sq.log 2>&1; tail -c 300 $O/rec_fetch.log; ls -la $O | tail -20"
