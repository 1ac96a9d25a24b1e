cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py -q -x > gpurun_out/check1.log 2>&1; echo "rc=$?" >> gpurun_out/check1.log
tail -3 gpurun_out/check1.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 | cut -c1-1500
