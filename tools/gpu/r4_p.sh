# round 4: drawing the next proposal ahead of the decision (MCD_MH_AHEAD_FROM), on / off over tree sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for args in "--sparse --dim 12" "--sparse --dim 62" "--sparse --dim 126" "--sparse --dim 254" "--sparse --dim 510" "--sparse --dim 1024" "--sparse --dim 2012" "--dim 598" "--dim 1024 --swap-period 2"; do
  for ah in 3 100000; do
    MCD_MH_AHEAD_FROM=$ah timeout -k 10 200 python bench.py --kind mh $args --chains 512 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ahead_from=$ah [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
  done
done | tee gpurun_out/r04/segment_ahead.txt
