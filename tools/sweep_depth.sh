# bench.py at the driver's step count for a few (parse lanes, key-ahead, depth) settings.  bash tools/sweep_depth.sh "16:10:4 24:12:4 ..." [extra bench flags]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in $1; do
  L=${cfg%%:*}; r=${cfg#*:}; K=${r%%:*}; D=${r#*:}
  ALFALFA_AMD_PARSE_LANES=$L timeout 400 python bench.py --steps 20 --warmup 2 --key-ahead $K --depth $D --no-cpu-baseline --no-verify --small-batches= --no-device-half $2 > gpurun_out/sweep_${L}_${K}_${D}.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/sweep_${L}_${K}_${D}.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("lanes $L K $K D $D value %.1fM" % (j["value"]/1e6), "steady %.1fM" % (j["steady_state"]["value"]/1e6), "mem", j["config"]["hbm_in_use_after_timed_region_gb"], "host", j["stages"]["host_prepass_and_staging_s_per_step"], j["timed_region"])
else: print("lanes $L K $K D $D failed"); print(open("gpurun_out/sweep_${L}_${K}_${D}.log").read()[-1500:])
PY
done
