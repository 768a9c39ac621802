# token kernel under load: inter-frame chains only, S streams x 11 frames in one launch.  bash tools/probe_load.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in 16:480 16:1120 16:2200 8:2200 24:2200 12:2200; do
  L=${cfg%%:*}; S=${cfg#*:}
  echo "lanes $L streams $S: $(ALFALFA_AMD_PARSE_LANES=$L timeout 300 python tools/parse_probe.py --streams $S --first 1 --reps 1 2>&1 | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
ALFALFA_AMD_PARSE_LANES=16 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/prof_load -o sq -- python $GRAFT_REPO_ROOT/tools/parse_probe.py --streams 2200 --first 1 --reps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_load.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_load.log
cd $GRAFT_REPO_ROOT && python tools/rocpd_pmc_summary.py gpurun_out/prof_load/sq_results.db 2>&1 | tail -30
