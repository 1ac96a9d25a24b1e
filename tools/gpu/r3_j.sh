# round 3: the streaming chain kernel at R = 6 and 8 (trees of 259 .. 514 nodes): tests, then lock steps against the two-launch path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "streaming or large_tree or incremental" > gpurun_out/r03/j_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03/j_tests.log
tail -6 gpurun_out/r03/j_tests.log
out=gpurun_out/r03/j_chain_big_r68.jsonl; : > $out
for pp in 0 1; do
  for cfg in "129 512" "136 512" "160 512" "193 512" "257 512" "257 64" "193 1024"; do
    set -- $cfg
    r=$(MCD_MH_PER_PHASE=$pp timeout -k 10 200 python tools/bench_mh_large.py $1 $2 4000 2>&1 | tail -1)
    echo "{\"per_phase\": $pp, \"r\": $r}" >> $out
  done
done
cat $out
