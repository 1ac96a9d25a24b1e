cd $GRAFT_REPO_ROOT
for opt in "" "--no-graph" "" "--no-graph" "--graph-chunk 5" "--graph-chunk 10"; do
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mh $opt 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('opt=[$opt] value %.1f M  ms_per_step %.2f us  kernel %.2f us' % (d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_us_per_launch']))
"
done
