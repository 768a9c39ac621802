#!/bin/bash
# Round 5, third GPU session: host lanes (AA_SUBMIT_HOST without blocking) -- tests, then the driver's command; 3 against 4 worker
# workgroups per CU and the expansion's stream over 20 steps.
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05c; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
timeout 400 python -m pytest tests/test_gpu_device_parse.py -q -m gpu -x --timeout 200 -k "host_lanes or key_frames_of_big or small_calls or info" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log | cut -c1-300
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "lanes busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "host waited", t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", d.get("verified_bit_exact_vs_reference"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   memory", d.get("memory"))
except Exception as ex: print("no line", ex)
PY
}
echo "== bench default (the driver's command)"; timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo rc=$?; line $O/bench_default.log; grep -i "error\|Traceback" $O/bench_default.err | head -5
B="python bench.py --steps 20 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0"
echo "== bench no urgent host"; timeout 300 $B --no-urgent-host > $O/bench_nourgent.log 2> $O/bench_nourgent.err; echo rc=$?; line $O/bench_nourgent.log
echo "== bench 4 wgs per CU"; ALFALFA_AMD_WGS_PER_CU=4 timeout 300 $B > $O/bench_wgs4.log 2> $O/bench_wgs4.err; echo rc=$?; line $O/bench_wgs4.log
echo "== bench expansion on compute"; ALFALFA_AMD_EXPAND_ON_COMPUTE=1 timeout 300 $B > $O/bench_expcompute.log 2> $O/bench_expcompute.err; echo rc=$?; line $O/bench_expcompute.log
