cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-mh"
pr() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(json.dumps({'args':'$1','form':d['config']['form'],'us':round(d['roofline']['kernel_us_per_launch'],2)}))
    except Exception as e: print('$1', 'ERR', l[:200])
"; }
for n in 160 200 256 384 512 768 1024; do for b in 16 64 256 512 1024; do for k in logpdf tree; do
MCD_SPLIT=1 $B --n $n --chains $b --kind $k 2>/dev/null | tail -1 | pr "split n=$n b=$b $k" | tee -a gpurun_out/r2_window.jsonl
MCD_SPLIT=0 $B --n $n --chains $b --kind $k 2>/dev/null | tail -1 | pr "sweep n=$n b=$b $k" | tee -a gpurun_out/r2_window.jsonl
done; done; done
