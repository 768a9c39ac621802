#!/bin/bash
# Round 5, first GPU session: the token lanes' new shape (three probability planes per lane, block ends deferred) against last
# round's library, built here as gpurun_in/libs/base.so -- device-parse tests, the entropy decode alone, the bench A/B.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05a; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
timeout 400 python -m pytest tests/test_gpu_device_parse.py tests/test_gpu_packed_coefficients.py tests/test_gpu_lane_per_partition.py -q -m gpu -x --timeout 200 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in new base k2 k8; do
  if [ $v = new ]; then unset ALFALFA_AMD_LIB; else export ALFALFA_AMD_LIB=$PWD/gpurun_in/libs/$v.so; fi
  echo "== probe $v"; timeout 200 python tools/parse_probe.py --streams 2200 --reps 1 > $O/probe_$v.log 2>&1; echo rc=$?; tail -1 $O/probe_$v.log | cut -c1-900
done
for v in new base; do
  if [ $v = new ]; then unset ALFALFA_AMD_LIB; else export ALFALFA_AMD_LIB=$PWD/gpurun_in/libs/$v.so; fi
  echo "== bench $v"; timeout 300 python bench.py --steps 12 --warmup 3 --secondary '' --small-batches '' --no-cpu-baseline --lanes-only-steps 0 > $O/bench_$v.log 2> $O/bench_$v.err; echo rc=$?
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$v.log") if l.startswith("{")][-1])
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "verified", d.get("verified") or d.get("bit_exact"))
    e=d.get("entropy_decode_roof") or {}
    print({k:e.get(k) for k in ("sustained_bools_per_s","lanes","us_per_step_lone_chain")}, (e.get("in_kernel_accounting") or {}))
except Exception as ex: print("no line", ex)
PY
done
