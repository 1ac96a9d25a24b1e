cd $GRAFT_REPO_ROOT
for B in 512 64; do
for pr in 0 4 8 16; do
   MCD_SPLIT=1 MCD_SPLIT_PROBE=$pr timeout -k 10 120 python bench.py --n 256 --chains $B --steps 5000 --warmup 500 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=256 B=$B probe=$pr kernel us %.2f' % (d['roofline']['kernel_us_per_launch']))
"
done
MCD_SPLIT=0 timeout -k 10 120 python bench.py --n 256 --chains $B --steps 5000 --warmup 500 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=256 B=$B sweep kernel us %.2f' % (d['roofline']['kernel_us_per_launch']))
"
done
