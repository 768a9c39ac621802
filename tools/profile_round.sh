# Round profile on the MI355X box: rocprofv3 kernel trace + HBM / SQ counters of bench.py, every counter set in its own pass
# (gpurun refuses --pmc together with the trace domains that crash nodes).  Usage: bash tools/profile_round.sh r02
# Writes gpurun_out/prof_<tag>_<config>/ (scratch); tools/profile_summary.py turns it into profiles/<tag>_<config>.md and
# profiles/pmc_traffic.json.  Counter passes serialise the kernels, so they run a shallow pipeline (2 steps ahead): traffic
# per macroblock does not depend on how deep the pipeline is.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=1080p_inter_lf
B="python $R/bench.py --config $C --steps 2 --warmup 0 --small-batches= --no-cpu-baseline --no-verify"
O=$R/gpurun_out/prof_${TAG}_$C
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O -o kt -- $B > $O/kt.log 2>&1
P="$B --key-ahead 2 --depth 2 --no-device-half"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $P > $O/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $P > $O/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O -o sq -- $P > $O/sq.log 2>&1
ls -la $O | tail -8
C=1080p_inter_lf_subpel
O=$R/gpurun_out/prof_${TAG}_$C
mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats -d $O -o kt -- python $R/bench.py --config $C --steps 4 --warmup 0 --small-batches= --no-cpu-baseline > $O/kt.log 2>&1
ls -la $O | tail -4
