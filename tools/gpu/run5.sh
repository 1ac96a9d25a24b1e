cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py -x -q -k "not stress" 2>&1 | tail -2
B="python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-mh"
pr() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', round(d['roofline']['kernel_us_per_launch'],2),'us')
    except Exception as e: print('$1', 'ERR', l[:200])
"; }
for n in 256 320 1024; do
for p in 0 8 16; do MCD_SPLIT_PROBE=$p $B --n $n 2>/dev/null | tail -1 | pr "n=$n probe=$p"; done
done
$B --n 256 --kind tree 2>/dev/null | tail -1 | pr "tree 255"
$B --n 256 --chains 64 2>/dev/null | tail -1 | pr "256x64"
$B --n 256 --chains 1024 2>/dev/null | tail -1 | pr "256x1024"
MCD_SPLIT_SCATTER=1 $B --n 256 2>/dev/null | tail -1 | pr "256x512 scatter"
python tools/microbench/split_stamps.py 256 512 2>&1 | grep -v amdgpu | head -12
