#!/bin/bash
# Round 5, eighteenth GPU session: top-up at an eighth is the default now -- the driver's command and 40 steps once more on the final tree.
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05r; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    print("   retired", t.get("worker_grids_retired"), "grids/wgs", t.get("worker_grids_launched"), t.get("worker_workgroups_launched"), "memory", (d.get("memory") or {}).get("hbm_taken_by_the_context_gb"), (d.get("memory") or {}).get("inside_the_budget"))
    print("   secondary", {k:(v or {}).get("value") for k,v in (d.get("secondary") or {}).items()}, "small64", ((d.get("small_batches") or {}).get("64") or {}).get("mb_per_s"), "delivery", (d.get("delivery") or {}).get("gb_per_s"))
except Exception as ex: print("no line", ex)
PY
}
timeout 240 python -m pytest tests/test_gpu_device_parse.py -m gpu -x -q > $O/gpu_tests_device_parse.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests_device_parse.log
echo "== the driver's command"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo rc=$?; line $O/bench_default.log; grep -i "Error" $O/bench_default.err | tail -2 | cut -c1-300
echo "== 40 steps"; timeout 400 python bench.py --steps 40 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half > $O/bench_40.log 2> $O/bench_40.err; echo rc=$?; line $O/bench_40.log; grep -i "Error" $O/bench_40.err | tail -2 | cut -c1-300
