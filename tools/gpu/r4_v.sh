# round 4: same-box A/B of two builds of the library (tools/microbench/libprev.so = the commit before the segment loop was rewritten) on the small-tree workloads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for rep in 1 2; do
for lib in "" tools/microbench/libprev.so; do
  for args in "--kind mh --sparse --dim 12 --chains 128" "--kind mh --sparse --dim 12 --chains 512" "--kind mh --dim 30 --chains 512" "--kind mh --dim 256 --chains 512"; do
  MCD_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lib [$lib] [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
  done
  MCD_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 300 python bench.py --kind e2e 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)['e2e']; print('lib [$lib] e2e wall', round(d['wall_s'], 2), 'burn-in', round(d['burn_in_s'], 2), 'run', round(d['run_s'], 2))" || exit 1
done
done | tee gpurun_out/r04/ab_prev.txt
