#!/bin/bash
# Round 4, sixth GPU session: default bench with the new fill order; A/B of what the reconstruction kernels find free beside the resident
# workers (loop-filter strips of 4 macroblocks = 11 KB of LDS instead of 19; three worker workgroups per CU instead of four).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
want=" ${*:-1 2 3} "
run() { case "$want" in *" $1 "*) shift; echo "== $*"; "$@";; esac; }
S="--steps 12 --warmup 3 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.1f M steady %.1f M first step %d ms" % (d["value"] / 1e6, d["steady_state"]["value"] / 1e6, d["timed_region"]["step_done_at_ms"][0]),
          "lanes", d["entropy_decode_roof"]["lanes"], "kernels beside workers", {k: v and v["avg_launch_us"] for k, v in d["kernels"].items() if k.startswith("recon") or k == "loopfilter"},
          "verified", d["verified_bit_exact_vs_reference"] and d["verified_bit_exact_vs_reference"]["bit_exact"])
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
run 1 bash -c 'timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/r04f_bench.log 2> gpurun_out/r04f_bench.err; echo rc=$?; grep "^\[bench" gpurun_out/r04f_bench.err | cut -c1-330'
if [[ "$want" == *" 2 "* ]]; then
  timeout 200 python bench.py $S > gpurun_out/r04f_ab_default.log 2> gpurun_out/r04f_ab_default.err; line gpurun_out/r04f_ab_default.log
  ALFALFA_AMD_LIB=$PWD/gpurun_in/libalfalfa_amd_strip4.so timeout 200 python bench.py $S > gpurun_out/r04f_ab_strip4.log 2> gpurun_out/r04f_ab_strip4.err; line gpurun_out/r04f_ab_strip4.log
  ALFALFA_AMD_WGS_PER_CU=3 timeout 200 python bench.py $S > gpurun_out/r04f_ab_wgs3.log 2> gpurun_out/r04f_ab_wgs3.err; line gpurun_out/r04f_ab_wgs3.log
  ALFALFA_AMD_WGS_PER_CU=3 ALFALFA_AMD_LIB=$PWD/gpurun_in/libalfalfa_amd_strip4.so timeout 200 python bench.py $S > gpurun_out/r04f_ab_wgs3_strip4.log 2> gpurun_out/r04f_ab_wgs3_strip4.err; line gpurun_out/r04f_ab_wgs3_strip4.log
fi
run 3 bash -c 'ALFALFA_AMD_LIB=$PWD/gpurun_in/libalfalfa_amd_strip4.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_1080p.py tests/test_reference_callers.py -q -m gpu --timeout 300 -k "not two_step" > gpurun_out/r04f_tests_strip4.log 2>&1; echo rc=$?; tail -4 gpurun_out/r04f_tests_strip4.log | cut -c1-300'
