cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py tests/test_gpu_parity.py -x -q -k "split or synthetic_logpdf" 2>&1 | tail -2
B="python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-mh"
pr() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', round(d['roofline']['kernel_us_per_launch'],2),'us')
    except Exception as e: print('$1', 'ERR', l[:200])
"; }
for a in "--n 256" "--n 256 --chains 1024" "--n 256 --chains 768" "--n 256 --chains 1024" "--n 1024" "--n 256 --kind tree" "--n 1024 --kind tree"; do
$B $a 2>/dev/null | tail -1 | pr "$a"
done
