cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02d; mkdir -p $O
P="python $R/tools/parse_probe.py --streams 16 --reps 1"
export ALFALFA_AMD_PARSE_LANES=8
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O -o sq -- $P > $O/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $O -o sq2 -- $P > $O/sq2.log 2>&1
ls $O
