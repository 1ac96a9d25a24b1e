cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_mh.py tests/test_gpu_sparse.py -q -m gpu -x -k "prior_waves or incremental or large_tree or segments or sparse_mh or native_sparse or twin" > gpurun_out/r04/tail_tests.log 2>&1 || { tail -30 gpurun_out/r04/tail_tests.log; exit 1; }
tail -2 gpurun_out/r04/tail_tests.log
for args in "--sparse --dim 12" "--sparse --dim 62" "--sparse --dim 126" "--sparse --dim 254" "--sparse --dim 1024" "--dim 598"; do
  for ah in 3 100000; do
    MCD_MH_AHEAD_FROM=$ah timeout -k 10 200 python bench.py --kind mh $args --chains 512 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ahead_from=$ah [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
  done
done | tee gpurun_out/r04/segment_ahead.txt
