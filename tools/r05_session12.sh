#!/bin/bash
# Round 5, twelfth GPU session: the reconstruction chain paces the plateau again (9.5 + 9.0 ms per call beside the workers against
# 3.1 + 2.4 alone): two worker workgroups per CU with as many lanes (two SIMDs of every CU free) against three.
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05l; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "lanes", e.get("lanes_per_workgroup"), e.get("workgroups_per_cu"), "busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    m=d.get("memory") or {}; print("   memory", {k:m.get(k) for k in ("pool_gb","coefficient_heap_mapped_gb","hbm_taken_by_the_context_gb","inside_the_budget")}, "pool_waits", t.get("pool_waits"), "grids/wgs", t.get("worker_grids_launched"), t.get("worker_workgroups_launched"))
    print("   kernels", {k:(v or {}).get("avg_launch_us") for k,v in (d.get("kernels") or {}).items()})
except Exception as ex: print("no line", ex)
PY
}
B="python bench.py --steps 24 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half --depth 7"
echo "== 2 x 49"; ALFALFA_AMD_WGS_PER_CU=2 ALFALFA_AMD_MAX_LANES=64 timeout 300 $B > $O/bench_2x49.log 2> $O/bench_2x49.err; echo rc=$?; line $O/bench_2x49.log; grep -i "Error" $O/bench_2x49.err | tail -2 | cut -c1-300
echo "== 2 x 40"; ALFALFA_AMD_WGS_PER_CU=2 ALFALFA_AMD_MAX_LANES=40 timeout 300 $B > $O/bench_2x40.log 2> $O/bench_2x40.err; echo rc=$?; line $O/bench_2x40.log; grep -i "Error" $O/bench_2x40.err | tail -2 | cut -c1-300
echo "== 3 x 37 (default)"; timeout 300 $B > $O/bench_3x37.log 2> $O/bench_3x37.err; echo rc=$?; line $O/bench_3x37.log; grep -i "Error" $O/bench_3x37.err | tail -2 | cut -c1-300
