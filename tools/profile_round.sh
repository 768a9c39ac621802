cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 2 --no-cpu-baseline --no-verify --no-profile-pass"
O=$R/gpurun_out/prof_r01j
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o kt -- $B > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $B > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O -o sq -- $B > $O/sq.log 2>&1
ls -la $O
cd $R && timeout 600 python bench.py > gpurun_out/bench_r01j.log 2>&1; tail -1 gpurun_out/bench_r01d.log | cut -c1-300
