#!/bin/bash
# Round 5, tenth GPU session: the per-step series without the call that synchronised every step with its own reconstruction;
# a 24-step run, look-ahead depths, then the driver's command.
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05j; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "lanes busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms")); print("   host", t.get("host_ms_per_step"), "info", t.get("host_ms_per_step_in_aa_ctx_get_info"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    m=d.get("memory") or {}; print("   memory", {k:m.get(k) for k in ("limit_gb","pool_gb","coefficient_heap_mapped_gb","hbm_taken_by_the_context_gb","inside_the_limit")}, "pool_waits", t.get("pool_waits"), "put off", t.get("hand_overs_put_off_for_lack_of_room"), t.get("of_which_refused_by_the_library_at_its_memory_limit"), "grids/wgs", t.get("worker_grids_launched"), t.get("worker_workgroups_launched"))
    print("   kernels", {k:(v or {}).get("avg_launch_us") for k,v in (d.get("kernels") or {}).items()})
    print("   delivery", {k:(d.get("delivery") or {}).get(k) for k in ("value","gb_per_s","between_fill_and_drain_gb_per_s","error")}); s=d.get("secondary") or {}
    for k,v in s.items(): print("   secondary", k, {x:v.get(x) for x in ("value","ms_per_step","error")}, "cpu", (v.get("cpu_baseline") or {}).get("value"), "lpp", {x:(v.get("with_a_lane_per_partition") or {}).get(x) for x in ("value","error")})
    print("   small", {k:v.get("mb_per_s") for k,v in (d.get("small_batches") or {}).items()}, "lanes only", (d.get("all_frames_on_gpu_lanes") or {}).get("value"), (d.get("all_frames_on_gpu_lanes") or {}).get("between_fill_and_drain_value"))
except Exception as ex: print("no line", ex)
PY
}
B="python bench.py --steps 24 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half"
for kd in "12 8" "11 5" "14 8"; do
  set -- $kd
  echo "== bench key-ahead $1 depth $2"; timeout 300 $B --key-ahead $1 --depth $2 > $O/bench_k$1_d$2.log 2> $O/bench_k$1_d$2.err; echo rc=$?; line $O/bench_k$1_d$2.log; grep -i "Error" $O/bench_k$1_d$2.err | tail -2 | cut -c1-300
done
echo "== bench default (the driver's command)"; timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo rc=$?; line $O/bench_default.log; grep -i "error\|Traceback" $O/bench_default.err | head -5
