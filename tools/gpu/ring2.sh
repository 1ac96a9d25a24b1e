cd $GRAFT_REPO_ROOT
for lib in "" tools/microbench/libsplit_ring8.so; do
 for n in 1024 768 512 384; do
  for k in logpdf tree grad; do
   MCD_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 120 python bench.py --n $n --kind $k --steps 3000 --warmup 300 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lib=${lib:-ring16} n=$n $k kernel us %.2f  wall %.2f' % (d['roofline']['kernel_us_per_launch'], d['ms_per_step']*1e3))
"
  done
 done
done
