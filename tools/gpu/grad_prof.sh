cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gp
export TMPDIR=/tmp
cd /tmp
for n in 1024 512 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gp/t$n -- python3 $GRAFT_REPO_ROOT/bench.py --n $n --kind tree_grad --steps 500 --warmup 50 --no-cpu-baseline --no-mh > /dev/null 2>&1
  echo "== n=$n"; cat $GRAFT_REPO_ROOT/gpurun_out/gp/t$n/*/*_kernel_stats.csv | cut -c1-200 | head -6
done
cd $GRAFT_REPO_ROOT
for n in 160 192 224 256; do
 for B in 16 64 512 1024; do
  for k in grad tree_grad; do
   for sp in 1 0; do
    MCD_SPLIT=$sp timeout -k 10 120 python bench.py --n $n --chains $B --kind $k --steps 2000 --warmup 200 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=$n B=$B $k split=$sp us/step %.2f' % (d['ms_per_step']*1e3))
"
   done
  done
 done
done
