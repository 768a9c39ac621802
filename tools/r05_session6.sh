#!/bin/bash
# Round 5, sixth GPU session: look-ahead depth against the memory budget (the steady state is paced by frames in flight), and the step
# with its LDS reads asked for a step ahead (probe).
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05f; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "lanes busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    m=d.get("memory") or {}; print("   memory", {k:m.get(k) for k in ("limit_gb","pool_gb","coefficient_heap_mapped_gb","hbm_taken_by_the_context_gb","inside_the_limit")}, "pool_waits", t.get("pool_waits"), "put off", t.get("hand_overs_put_off_for_lack_of_room"), t.get("of_which_refused_by_the_library_at_its_memory_limit"), "starved/evicted/handed back", (d.get("kernel_stats") or {}).get("lanes_starved"), t.get("frames_evicted"), t.get("frames_handed_back_for_lack_of_memory"))
except Exception as ex: print("no line", ex)
PY
}
for v in new pre1; do
  if [ $v = new ]; then unset ALFALFA_AMD_LIB; else export ALFALFA_AMD_LIB=$R/gpurun_in/libs/$v.so; fi
  echo "== probe $v"; timeout 200 python tools/parse_probe.py --streams 2200 --reps 1 > $O/probe_$v.log 2>&1; echo rc=$?; tail -1 $O/probe_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['profile']; print(d['parse_wall_s'], p['wave_seconds'], p['us_per_wave_step'], p['wave_steps'], p['frac_steps'], p['us_per_boundary_pass'])"
done
unset ALFALFA_AMD_LIB
B="python bench.py --steps 24 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half"
for kd in "12 8" "11 5" "10 4" "12 6"; do
  set -- $kd
  echo "== bench key-ahead $1 depth $2"; timeout 300 $B --key-ahead $1 --depth $2 > $O/bench_k$1_d$2.log 2> $O/bench_k$1_d$2.err; echo rc=$?; line $O/bench_k$1_d$2.log; grep -i "Error" $O/bench_k$1_d$2.err | tail -2 | cut -c1-300
done
