#!/bin/bash
# Round 5, seventeenth GPU session: 40 steps on the final tree (do the top-ups keep the lanes up once every worker stream has had a grid?);
# a bigger share of the plateau's key frames on the host lanes (tokens and reconstruction share the CUs; the host's cores are idle there).
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05q; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    print("   retired", t.get("worker_grids_retired"), "host", t.get("host_ms_per_step"), "host frames", t.get("frames_parsed_on_host_cores"), "grids/wgs", t.get("worker_grids_launched"), t.get("worker_workgroups_launched"), "threads", (d.get("config") or {}).get("host_threads"))
except Exception as ex: print("no line", ex)
PY
}
B="python bench.py --steps 20 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half"
echo "== 40 steps"; timeout 400 $B --steps 40 > $O/bench_40.log 2> $O/bench_40.err; echo rc=$?; line $O/bench_40.log; grep -i "Error" $O/bench_40.err | tail -2 | cut -c1-300
echo "== host share 250 ms"; timeout 300 $B --host-share-ms 250 > $O/bench_share250.log 2> $O/bench_share250.err; echo rc=$?; line $O/bench_share250.log; grep -i "Error" $O/bench_share250.err | tail -2 | cut -c1-300
