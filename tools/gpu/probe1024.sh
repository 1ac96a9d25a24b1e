cd $GRAFT_REPO_ROOT
for n in 1024 512; do
for pr in 0 4 8 16; do
   MCD_SPLIT_PROBE=$pr timeout -k 10 120 python bench.py --n $n --steps 3000 --warmup 300 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=$n probe=$pr kernel us %.2f' % (d['roofline']['kernel_us_per_launch']))
"
done
done
