cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests/test_gpu_split.py -q -x --durations=5 2>&1 | tail -8
