# round 4: the prior wave drawing the next proposal, with TUNED proposals (bench.py --tune-periods: the reference samples after burn-in with auto-tuning)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for tp in 20 0; do
for pd in 1 0; do
  for args in "--kind mh --sparse --dim 12 --chains 128" "--kind mh --sparse --dim 1024 --chains 512" "--kind mh --sparse --dim 2012 --chains 512" "--kind mh --dim 598 --chains 512" "--kind mh --dim 1024 --chains 512 --swap-period 2" "--kind mh --dim 256 --chains 512"; do
  MCD_MH_PRIOR_DRAWS=$pd timeout -k 10 200 python bench.py $args --tune-periods $tp --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tune_periods=$tp prior_draws=$pd [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step, acceptance', round(d['mh']['acceptance_rate'], 3))" || exit 1
  done
done; done | tee gpurun_out/r04/prior_draws_tuned_ab.txt
