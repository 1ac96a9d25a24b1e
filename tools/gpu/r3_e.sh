# round 3: fast path of the incremental evaluation (listed slots): tests, timings, stamps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "streaming or large_tree or lockstep_parity or cpp" > gpurun_out/r03/e_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03/e_tests.log
tail -5 gpurun_out/r03/e_tests.log
out=gpurun_out/r03/e_incremental.jsonl; : > $out
for cfg in "40 512" "70 512" "100 512" "129 64" "129 512" "129 1024"; do
  set -- $cfg
  r=$(timeout -k 10 120 python tools/bench_mh_large.py $1 $2 8000 2>&1 | tail -1)
  echo "{\"incremental\": 1, \"r\": $r}" >> $out
done
cat $out
timeout -k 10 120 python tools/microbench/mhbig_stamps.py 256 512 8000 > gpurun_out/r03/e_mhbig_stamps.txt 2>&1
cat gpurun_out/r03/e_mhbig_stamps.txt
