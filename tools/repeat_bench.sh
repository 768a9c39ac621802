# bench.py at the driver's settings under a few settings: bash tools/repeat_bench.sh "name:ENV=V,ENV=V[:bench flags with + for space] name2:..."
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for item in $1; do
  name=${item%%:*}; rest=${item#*:}; envs=$(echo ${rest%%:*} | tr ',' ' '); flags=""; case "$rest" in *:*) flags=$(echo ${rest#*:} | tr '+' ' ');; esac
  env $envs timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --small-batches= --no-device-half $flags > gpurun_out/rep_$name.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/rep_$name.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); t=j["timed_region"]; print("$name value %.1fM" % (j["value"]/1e6), "mem", j["config"]["hbm_in_use_after_timed_region_gb"], "decode-call", t["host_ms_per_step"]["decode_batch_calls_incl_wait_for_parse"], "parse-wait", t["host_waited_for_parse_ms_per_step"], "compute-wait", t["host_waited_for_compute_stream_ms_per_step"], "alloc", t["host_in_pool_allocator_ms_per_step"], t["slab_mallocs"], t["step_done_at_ms"])
else: print("$name failed", open("gpurun_out/rep_$name.log").read()[-800:])
PY
done
