#!/bin/bash
# Round 5, fourth GPU session: the GPU suite on the tree with host lanes for the hybrid share, the hard memory limit and two
# urgent groups; then the driver's command, twice; counters of the token workers on this round's kernel.
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05d; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
timeout 700 python -m pytest tests -q -m gpu -x --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log | cut -c1-300
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}; t=d.get("timed_region") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "lanes busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "waits parse/compute", t.get("host_waited_for_parse_ms_per_step"), t.get("host_waited_for_compute_stream_ms_per_step"), "bit-exact", (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"))
    print("   step_done", t.get("step_done_at_ms"))
    print("   per_step", (t.get("per_step") or {}).get("series"))
    m=d.get("memory") or {}; print("   memory", {k:m.get(k) for k in ("limit_gb","pool_gb","coefficient_heap_mapped_gb","hbm_taken_by_the_context_gb")}, "pool_waits", t.get("pool_waits"))
except Exception as ex: print("no line", ex)
PY
}
echo "== bench default (the driver's command)"; timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo rc=$?; line $O/bench_default.log; grep -i "error\|Traceback" $O/bench_default.err | head -5
B="python bench.py --steps 20 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0"
echo "== bench again, no extras"; timeout 300 $B > $O/bench_again.log 2> $O/bench_again.err; echo rc=$?; line $O/bench_again.log
echo "== bench host share 0"; timeout 300 $B --host-share-ms 0 > $O/bench_share0.log 2> $O/bench_share0.err; echo rc=$?; line $O/bench_share0.log
echo "== bench 40 steps"; timeout 400 python bench.py --steps 40 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 > $O/bench_40.log 2> $O/bench_40.err; echo rc=$?; line $O/bench_40.log
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/parse_probe.py --streams 96 --frames 12 --reps 1"
export ALFALFA_AMD_WORKER_LINGER_MS=0 ALFALFA_AMD_ROUTE=device
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O -o tok_$n -- $P > $O/tok_$n.log 2>&1; echo "pmc $n rc=$?"; grep "^{" $O/tok_$n.log | tail -1 | cut -c1-700
done
