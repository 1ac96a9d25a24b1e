cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r2_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_gpu_tests.log
tail -30 gpurun_out/r2_gpu_tests.log
