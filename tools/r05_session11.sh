#!/bin/bash
# Round 5, eleventh GPU session: look-ahead depths 12/6 and 12/7 on the fixed bench; the lane-per-partition leg after the heap estimate
# fix (in the default command, --secondary only that config); rocprofv3 kernel trace of a shortened driver's command.
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05k; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "lanes busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms")); print("   host", t.get("host_ms_per_step"), "info", t.get("host_ms_per_step_in_aa_ctx_get_info"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    m=d.get("memory") or {}; print("   memory", {k:m.get(k) for k in ("limit_gb","pool_gb","coefficient_heap_mapped_gb","hbm_taken_by_the_context_gb","inside_the_budget")}, "pool_waits", t.get("pool_waits"), "put off", t.get("hand_overs_put_off_for_lack_of_room"), t.get("of_which_refused_by_the_library_at_its_memory_limit"), "grids/wgs", t.get("worker_grids_launched"), t.get("worker_workgroups_launched"))
    s=d.get("secondary") or {}
    for k,v in s.items(): print("   secondary", k, {x:v.get(x) for x in ("value","ms_per_step","error","frames_handed_back_for_lack_of_memory","heap_grows")}, "lpp", {x:(v.get("with_a_lane_per_partition") or {}).get(x) for x in ("value","ms_per_step","error","frames_handed_back_for_lack_of_memory","heap_grows","host_waited_for_parse_ms_per_step","lone_key_frame_parse_s")})
except Exception as ex: print("no line", ex)
PY
}
B="python bench.py --steps 24 --warmup 5 --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half"
echo "== bench key-ahead 12 depth 6 (+ the sub-pel config with and without a lane per partition)"; timeout 400 $B --key-ahead 12 --depth 6 --secondary 1080p_inter_lf_subpel > $O/bench_k12_d6.log 2> $O/bench_k12_d6.err; echo rc=$?; line $O/bench_k12_d6.log; grep -i "Error" $O/bench_k12_d6.err | tail -2 | cut -c1-300
echo "== bench key-ahead 12 depth 7"; timeout 300 $B --key-ahead 12 --depth 7 --secondary= > $O/bench_k12_d7.log 2> $O/bench_k12_d7.err; echo rc=$?; line $O/bench_k12_d7.log; grep -i "Error" $O/bench_k12_d7.err | tail -2 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel trace of the driver's command, shortened"
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o kt -- python $R/bench.py --steps 8 --warmup 2 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 > $O/kt.log 2> $O/kt.err; echo rc=$?; ls -la $O | grep kt | head; line $O/kt.log
