# round 4: dense proposals proposed by the preceding segment's launch (MCD_MH_SEG_TAIL): identical chains, then A/B of the three MH workloads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_mh.py tests/test_gpu_sparse.py -q -m gpu -x -k "prior_waves or incremental or large_tree or segments or sparse_mh or native_sparse" > gpurun_out/r04/tail_tests.log 2>&1 || { tail -30 gpurun_out/r04/tail_tests.log; exit 1; }
tail -3 gpurun_out/r04/tail_tests.log
for tailv in 1 0; do
  for args in "--dim 1024 --chains 512 --swap-period 2" "--sparse --dim 1024 --chains 512" "--sparse --dim 2012 --chains 512" "--dim 598 --chains 512"; do
    MCD_MH_SEG_TAIL=$tailv timeout -k 10 200 python bench.py --kind mh $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tail=$tailv [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step', d.get('config', {}).get('path'))" || exit 1
  done
done
