# round 4: same-box A/B of the tree's library against tools/microbench/libprev.so (the last commit's build), segment workloads with tuned proposals
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "prior_waves or dense_proposals_at or incremental" > gpurun_out/r04/ae_quick.log 2>&1 || { tail -30 gpurun_out/r04/ae_quick.log; exit 1; }
tail -2 gpurun_out/r04/ae_quick.log
for rep in 1 2; do
for lib in "" tools/microbench/libprev.so; do
  for args in "--kind mh --sparse --dim 12 --chains 128" "--kind mh --sparse --dim 1024 --chains 512" "--kind mh --sparse --dim 2012 --chains 512" "--kind mh --dim 598 --chains 512" "--kind mh --dim 1024 --chains 512 --swap-period 2"; do
  MCD_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python bench.py $args --tune-periods 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lib [$lib] [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
  done
done; done | tee gpurun_out/r04/ab_prev2.txt
