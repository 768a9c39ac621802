#!/bin/bash
# Round 5, second GPU session: formulations of the token step (adv select, stream byte asked for a step ahead), the shader clock
# under this load, the expansion kernels on the utility stream (A/B), worker shapes, and -- item 3 of the review -- counters of
# the reconstruction / expansion kernels from a replay with no worker grid resident.
set -u
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed_coefficients.py -q -m gpu -x --timeout 200 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
( for i in $(seq 1 40); do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2 | tr '\n' ' '; echo; sleep 1; done > $O/clocks.log 2>&1 ) &
for v in new v1 v2 v3; do
  if [ $v = new ]; then unset ALFALFA_AMD_LIB; else export ALFALFA_AMD_LIB=$R/gpurun_in/libs/$v.so; fi
  echo "== probe $v"; timeout 200 python tools/parse_probe.py --streams 2200 --reps 1 > $O/probe_$v.log 2>&1; echo rc=$?; tail -1 $O/probe_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['profile']; print(d['parse_wall_s'], p['wave_seconds'], p['us_per_wave_step'], p['wave_steps'], p['frac_steps'], p['us_per_boundary_pass'])"
done
wait
unset ALFALFA_AMD_LIB
B="python bench.py --steps 12 --warmup 3 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    e=d.get("entropy_decode_roof") or {}; a=e.get("in_kernel_accounting") or {}
    print({k:d.get(k) for k in ("value","ms_per_step")}, "steady", (d.get("steady_state") or {}).get("value"), "bools/s", e.get("sustained_bools_per_s"), "lanes busy", a.get("lanes_with_frame_per_period"), "us/step", a.get("us_per_wave_step"), "host waited", (d.get("timed_region") or {}).get("host_waited_for_compute_stream_ms_per_step"))
except Exception as ex: print("no line", ex)
PY
}
echo "== bench new (expansion on the utility stream)"; timeout 300 $B > $O/bench_new.log 2> $O/bench_new.err; echo rc=$?; line $O/bench_new.log
echo "== bench expansion on the compute stream"; ALFALFA_AMD_EXPAND_ON_COMPUTE=1 timeout 300 $B > $O/bench_expcompute.log 2> $O/bench_expcompute.err; echo rc=$?; line $O/bench_expcompute.log
echo "== bench 3 x 38 lanes"; ALFALFA_AMD_WGS_PER_CU=3 ALFALFA_AMD_MAX_LANES=40 timeout 300 $B > $O/bench_wgs3.log 2> $O/bench_wgs3.err; echo rc=$?; line $O/bench_wgs3.log
# ---- counters: every set its own pass, no trace domains beside --pmc except the kernel trace ----
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/recon_replay.py --streams 120 --frames 4 --reps 2"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 240 rocprofv3 --kernel-trace --pmc $c -d $O -o rec_$n -- $P > $O/rec_$n.log 2>&1; echo "pmc $n rc=$?"; tail -1 $O/rec_$n.log | cut -c1-600
done
ls -la $O | head -40
