# Round profile on the MI355X box: rocprofv3 kernel trace + HBM / SQ counters of bench.py, every counter set in its own pass
# (gpurun refuses --pmc together with the trace domains that crash nodes).  Usage: bash tools/profile_round.sh r02 [config ...]
# Writes gpurun_out/prof_<tag>_<config>/ (scratch) ; tools/profile_summary.py turns it into profiles/<tag>_<config>.md and
# profiles/pmc_traffic.json.
TAG=${1:-r02}; shift
CONFIGS=${@:-1080p_inter_lf}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in $CONFIGS; do
  B="python $R/bench.py --config $C --steps 2 --warmup 0 --small-batches= --no-cpu-baseline --no-verify"
  O=$R/gpurun_out/prof_${TAG}_$C
  mkdir -p $O
  timeout 400 rocprofv3 --kernel-trace --stats -d $O -o kt -- $B > $O/kt.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $B > $O/fetch.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $B > $O/write.log 2>&1
  if [ "$C" = "1080p_inter_lf" ]; then
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O -o sq -- $B > $O/sq.log 2>&1
  fi
  ls -la $O | tail -8
done
