cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/window2.jsonl
for n in 200 224 256; do
 for B in 32 64 128 256; do
  for k in logpdf tree; do
   for sp in 1 0; do
   MCD_SPLIT=$sp timeout -k 10 120 python bench.py --n $n --chains $B --kind $k --steps 5000 --warmup 500 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = {'n': $n, 'chains': $B, 'kind': '$k', 'split': $sp, 'kernel_us': round(d['roofline']['kernel_us_per_launch'], 3)}; print(json.dumps(r))
" | tee -a gpurun_out/window2.jsonl
   done
  done
 done
done
