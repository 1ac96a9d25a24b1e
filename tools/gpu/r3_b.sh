# round 3: sampler kernels compiled without machine LICM (no hoisted literals, no scratch) against the round-2 build
# (tools/microbench/libbase_licm.so, loaded through MCD_LIB_PATH): lock steps and leapfrog steps at the usual sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
out=gpurun_out/r03/b_licm.jsonl; : > $out
for lib in new base; do
  if [ $lib = base ]; then export MCD_LIB_PATH=$PWD/tools/microbench/libbase_licm.so; else unset MCD_LIB_PATH; fi
  for cfg in "12 64" "12 4096" "25 512" "70 512" "129 512" "129 1024" "129 4096" "200 512" "513 512"; do
    set -- $cfg
    r=$(timeout -k 10 120 python tools/bench_mh_large.py $1 $2 3000 2>&1 | tail -1)
    echo "{\"lib\": \"$lib\", \"tool\": \"mh\", \"r\": $r}" >> $out
  done
  for cfg in "12 64" "128 512" "128 4096" "512 512"; do
    set -- $cfg
    r=$(timeout -k 10 120 python tools/bench_leapfrog.py $1 $2 200 2>&1 | tail -1)
    echo "{\"lib\": \"$lib\", \"tool\": \"leapfrog\", \"r\": $r}" >> $out
  done
done
unset MCD_LIB_PATH
cat $out
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py tests/test_gpu_nuts.py -q -m gpu -x -k "streaming or chain_kernel_equals or lockstep_parity or workgroup_per_chain or prior_beside or large_tree or mc3 or nuts_follows or agree_with" > gpurun_out/r03/b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03/b_tests.log
tail -5 gpurun_out/r03/b_tests.log
