#!/bin/bash
# Round 5, nineteenth GPU session (the last minutes): the bench's per-rank sizing of host lanes / urgent groups on the final tree --
# the bench smoke tests (two ranks on one GPU among them), a short run of the driver's configuration, smoke().
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05s; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=60
timeout 150 python -m pytest tests/test_bench_smoke.py -m gpu -x -q > $O/gpu_tests_bench_smoke.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests_bench_smoke.log
timeout 100 python bench.py --steps 10 --warmup 3 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half > $O/bench_10.log 2> $O/bench_10.err; echo "bench rc=$?"
python - $O/bench_10.log <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); t=d["timed_region"]
    print({k:d.get(k) for k in ("value","ms_per_step")}, "bit-exact", d["verified_bit_exact_vs_reference"]["bit_exact"], "host_share", {k:d["host_share"].get(k) for k in ("host_lanes_of_this_rank","urgent_groups_planned","a_group_of_key_frames_on_the_host_route_would_take_ms","groups_whose_key_frames_took_the_host_route_in_the_timed_region")})
    print("   step_done", t["step_done_at_ms"])
except Exception as ex: print("no line", ex)
PY
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
