cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_device_parse.py -x -q > gpurun_out/r02g_tests.log 2>&1; tail -3 gpurun_out/r02g_tests.log
for L in 16 12 24 8; do
  ALFALFA_AMD_PARSE_LANES=$L timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-verify --small-batches= --no-device-half > gpurun_out/r02g_lanes$L.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/r02g_lanes$L.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("lanes $L value", j["value"], "steady", j["steady_state"], "tok", j["kernels"]["parse_tokens"]["ms_per_step"], j["stages"])
else: print("lanes $L failed"); print(open("gpurun_out/r02g_lanes$L.log").read()[-2000:])
PY
done
