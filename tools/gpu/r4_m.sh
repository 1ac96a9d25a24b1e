# round 4: A/B of one change on the dense segment workloads (values of MCD_AB passed through to the library's knobs by the caller)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "prior_waves or incremental or large_tree or segments" > gpurun_out/r04/ab_tests.log 2>&1 || { tail -30 gpurun_out/r04/ab_tests.log; exit 1; }
tail -2 gpurun_out/r04/ab_tests.log
for rep in 1 2; do
  for args in "--dim 1024 --chains 512 --swap-period 2" "--dim 598 --chains 512" "--dim 766 --chains 512"; do
    timeout -k 10 200 python bench.py --kind mh $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
  done
done
