#!/bin/bash
# round 3 evidence: the GPU test suite, the default bench line as the driver runs it, rocprofv3 kernel trace + HBM counter passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r03_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03_gpu_tests.log
tail -4 gpurun_out/r03_gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.log 2>&1
echo "rc=$?" >> gpurun_out/r03_bench_default.log
tail -c 600 gpurun_out/r03_bench_default.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=1080p_inter_lf
O=$R/gpurun_out/prof_r03_$C
mkdir -p $O
B="python $R/bench.py --config $C --steps 4 --warmup 0 --small-batches= --no-cpu-baseline --no-verify"
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o kt -- $B > $O/kt.log 2>&1
export ALFALFA_AMD_WORKER_LINGER_MS=0
P="python $R/bench.py --config $C --streams 120 --steps 2 --warmup 0 --small-batches= --no-cpu-baseline --no-verify --key-ahead 2 --depth 2 --no-device-half"
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $P > $O/fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $P > $O/write.log 2>&1
ls -la $O | tail -12
du -sh $O
