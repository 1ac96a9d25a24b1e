# round 4, closing pass: the whole -m gpu suite, smoke(), the README's Metropolis-Hastings figures
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=8 -x > gpurun_out/r04/gpu_tests.log 2>&1; rc=$?; tail -14 gpurun_out/r04/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 || exit 1
bash tools/gpu/r4_final.sh
