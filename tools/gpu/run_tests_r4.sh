# the whole -m gpu suite with durations (round 4)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1150 python -m pytest tests -q -m gpu --durations=12 -x > gpurun_out/r04/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04/gpu_tests.log
tail -25 gpurun_out/r04/gpu_tests.log
