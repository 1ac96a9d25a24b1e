# round 3: incremental likelihood of the large-tree two-launch path: tests, then lock steps with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 700 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "incremental or large_tree or workgroup_per_chain or prior_beside" > gpurun_out/r03/i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03/i_tests.log
tail -6 gpurun_out/r03/i_tests.log
out=gpurun_out/r03/i_inc_large.jsonl; : > $out
for inc in 1 0; do
  for cfg in "193 512" "257 512" "513 512" "513 64" "513 1024"; do
    set -- $cfg
    r=$(MCD_MH_INCREMENTAL=$inc timeout -k 10 200 python tools/bench_mh_large.py $1 $2 3000 2>&1 | tail -1)
    echo "{\"incremental\": $inc, \"r\": $r}" >> $out
  done
done
cat $out
