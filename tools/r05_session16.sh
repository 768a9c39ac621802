#!/bin/bash
# Round 5, sixteenth GPU session: the GPU suite on the final tree; how many workgroups must be gone before a top-up grid is launched
# (an eighth / a sixteenth of the GPU's instead of a quarter: session 15 shows jobs waiting while 65 of 768 workgroups are missing);
# the driver's command; counters of the token workers and of the header kernel as they are now (tables in LDS since session 8).
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05p; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
export ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    print("   retired", t.get("worker_grids_retired"), "host", t.get("host_ms_per_step"), "host frames", t.get("frames_parsed_on_host_cores"), "grids/wgs", t.get("worker_grids_launched"), t.get("worker_workgroups_launched"))
    m=d.get("memory") or {}; print("   memory", {k:m.get(k) for k in ("hbm_taken_by_the_context_gb","inside_the_budget")}, "roofline", d.get("roofline"))
    print("   secondary", {k:(v or {}).get("value") for k,v in (d.get("secondary") or {}).items()}, "small", d.get("small_batches"))
    print("   delivery", d.get("delivery")); print("   lpp", d.get("lane_per_partition")); print("   cpu", d.get("cpu_baseline"))
except Exception as ex: print("no line", ex)
PY
}
B="python bench.py --steps 20 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half"
echo "== top-up at a sixteenth"; ALFALFA_AMD_TOPUP_DIV=16 timeout 300 $B > $O/bench_topup16.log 2> $O/bench_topup16.err; echo rc=$?; line $O/bench_topup16.log; grep -i "Error" $O/bench_topup16.err | tail -2 | cut -c1-300
echo "== top-up at an eighth"; ALFALFA_AMD_TOPUP_DIV=8 timeout 300 $B > $O/bench_topup8.log 2> $O/bench_topup8.err; echo rc=$?; line $O/bench_topup8.log; grep -i "Error" $O/bench_topup8.err | tail -2 | cut -c1-300
echo "== the driver's command"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo rc=$?; line $O/bench_default.log; tail -12 $O/bench_default.err | cut -c1-400
# ---- counters: every set its own pass, no trace domains beside --pmc except the kernel trace ----
export ALFALFA_AMD_WORKER_LINGER_MS=0 ALFALFA_AMD_ROUTE=device
unset ALFALFA_AMD_TOKEN_PROFILE
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/parse_probe.py --streams 96 --frames 12 --reps 1"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o tok_$n -- $P > $O/tok_$n.log 2>&1; echo "pmc $n rc=$?"; grep "^{" $O/tok_$n.log | tail -1 | cut -c1-400
done
ls $O | head -40
