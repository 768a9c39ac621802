#!/bin/bash
# Round 5, twentieth GPU session (the last two minutes): where the host's time inside aa_decode_batch goes (ALFALFA_AMD_DECODE_TIMING=1).
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r05t; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=60 ALFALFA_AMD_DECODE_TIMING=1
timeout 110 python bench.py --steps 16 --warmup 3 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half > $O/bench_16.log 2> $O/bench_16.err; echo "bench rc=$?"
grep "aa_decode_batch host time" $O/bench_16.err | cut -c1-900
python - $O/bench_16.log <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); t=d["timed_region"]
    print({k:d.get(k) for k in ("value","ms_per_step")}, "bit-exact", d["verified_bit_exact_vs_reference"]["bit_exact"], "host", t["host_ms_per_step"], "waits", t["host_waited_for_parse_ms_per_step"], t["host_waited_for_compute_stream_ms_per_step"])
    print("   per_step", t["per_step"]["series"])
except Exception as ex: print("no line", ex)
PY
