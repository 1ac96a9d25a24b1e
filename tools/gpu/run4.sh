cd $GRAFT_REPO_ROOT
B="python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-mh"
pr() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', round(d['roofline']['kernel_us_per_launch'],2),'us')
    except Exception as e: print('$1', 'ERR', l[:200])
"; }
for n in 256 1024; do
for p in 0 4 8 16; do MCD_SPLIT_PROBE=$p $B --n $n 2>/dev/null | tail -1 | pr "n=$n probe=$p"; done
MCD_LIB_PATH=$PWD/tools/microbench/libsplit_v1.so $B --n $n 2>/dev/null | tail -1 | pr "n=$n kernarg-preload"
done
