# round 4: the next proposal drawn for both outcomes (chain wave: accepted, prior wave: rejected): tests, then A/B with tuned and untuned proposals
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 120 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "prior_waves and MCD_MH_PRIOR_DRAWS" > gpurun_out/r04/pd_quick.log 2>&1 || { tail -30 gpurun_out/r04/pd_quick.log; exit 1; }
tail -2 gpurun_out/r04/pd_quick.log
timeout -k 10 700 python -m pytest tests/test_gpu_mh.py tests/test_gpu_sparse.py -q -m gpu -x > gpurun_out/r04/mh_tests.log 2>&1 || { tail -30 gpurun_out/r04/mh_tests.log; exit 1; }
tail -2 gpurun_out/r04/mh_tests.log
for tp in 20 0; do
for pd in 1 0; do
  for args in "--kind mh --sparse --dim 12 --chains 128" "--kind mh --sparse --dim 126 --chains 512" "--kind mh --sparse --dim 1024 --chains 512" "--kind mh --sparse --dim 2012 --chains 512" "--kind mh --dim 598 --chains 512" "--kind mh --dim 1024 --chains 512 --swap-period 2"; do
  MCD_MH_PRIOR_DRAWS=$pd timeout -k 10 200 python bench.py $args --tune-periods $tp --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tune_periods=$tp prior_draws=$pd [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step, acceptance', round(d['mh']['acceptance_rate'], 3))" || exit 1
  done
done; done | tee gpurun_out/r04/prior_draws_tuned_ab.txt
for pd in 1 0; do
  MCD_MH_PRIOR_DRAWS=$pd timeout -k 10 300 python bench.py --kind e2e 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)['e2e']; print('prior_draws=$pd e2e wall', round(d['wall_s'], 2), 'burn-in', round(d['burn_in_s'], 2), 'run', round(d['run_s'], 2))" || exit 1
done | tee -a gpurun_out/r04/prior_draws_tuned_ab.txt
