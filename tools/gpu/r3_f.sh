cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 120 python tools/microbench/mhbig_stamps.py 256 512 8000 > gpurun_out/r03/f_mhbig_stamps.txt 2>&1
cat gpurun_out/r03/f_mhbig_stamps.txt
timeout -k 10 120 python tools/bench_mh_large.py 129 512 8000 2>&1 | tail -1
