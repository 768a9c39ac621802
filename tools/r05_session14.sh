#!/bin/bash
# Round 5, fourteenth GPU session: how long an idle worker wave should stay (session 13: 200 ms instead of 2 s -> 114.6 M against 97.6-103.1 M:
# the waves that leave give their CU's LDS and SIMD to the reconstruction kernels, in the plateau and above all in the drain).
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05n; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    print("   host", t.get("host_ms_per_step"), "host frames", t.get("frames_parsed_on_host_cores"), "grids/wgs", t.get("worker_grids_launched"), t.get("worker_workgroups_launched"), "threads", (d.get("config") or {}).get("host_threads"))
except Exception as ex: print("no line", ex)
PY
}
B="python bench.py --steps 20 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half"
echo "== linger 100 ms"; ALFALFA_AMD_WORKER_LINGER_MS=100 timeout 300 $B > $O/bench_linger100.log 2> $O/bench_linger100.err; echo rc=$?; line $O/bench_linger100.log; grep -i "Error" $O/bench_linger100.err | tail -2 | cut -c1-300
echo "== linger 50 ms"; ALFALFA_AMD_WORKER_LINGER_MS=50 timeout 300 $B > $O/bench_linger50.log 2> $O/bench_linger50.err; echo rc=$?; line $O/bench_linger50.log; grep -i "Error" $O/bench_linger50.err | tail -2 | cut -c1-300
echo "== linger 20 ms"; ALFALFA_AMD_WORKER_LINGER_MS=20 timeout 300 $B > $O/bench_linger20.log 2> $O/bench_linger20.err; echo rc=$?; line $O/bench_linger20.log; grep -i "Error" $O/bench_linger20.err | tail -2 | cut -c1-300
echo "== linger 200 ms"; ALFALFA_AMD_WORKER_LINGER_MS=200 timeout 300 $B > $O/bench_linger200.log 2> $O/bench_linger200.err; echo rc=$?; line $O/bench_linger200.log; grep -i "Error" $O/bench_linger200.err | tail -2 | cut -c1-300
