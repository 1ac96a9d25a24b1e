# round 3: incremental evaluation of sparse proposals in the streaming chain kernel -- parity tests, then lock-step timings with
# and without it (MCD_MH_INCREMENTAL=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "streaming or large_tree or mc3 or lockstep_parity" > gpurun_out/r03/c_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03/c_tests.log
tail -5 gpurun_out/r03/c_tests.log
out=gpurun_out/r03/c_incremental.jsonl; : > $out
for inc in 1 0; do
  for cfg in "40 512" "70 512" "100 512" "129 64" "129 512" "129 1024"; do
    set -- $cfg
    r=$(MCD_MH_INCREMENTAL=$inc timeout -k 10 120 python tools/bench_mh_large.py $1 $2 8000 2>&1 | tail -1)
    echo "{\"incremental\": $inc, \"r\": $r}" >> $out
  done
done
cat $out
