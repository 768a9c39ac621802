# What the round's numbers come from, in one gpurun call: the GPU test suite, the profiles, the default bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_tests.log 2>&1; tail -3 gpurun_out/r02_tests.log
bash tools/profile_round.sh r02 > gpurun_out/r02_profile.log 2>&1; tail -5 gpurun_out/r02_profile.log
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.log 2> gpurun_out/r02_bench_default.err; tail -c 600 gpurun_out/r02_bench_default.log
