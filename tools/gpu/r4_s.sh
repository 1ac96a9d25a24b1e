cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for args in "--sparse --dim 12" "--sparse --dim 1024" "--sparse --dim 2012" "--dim 598" "--dim 1024 --swap-period 2"; do
    timeout -k 10 200 python bench.py --kind mh $args --chains 512 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
done
for n in 12 1024; do
echo "== new loop, default, $n"; timeout -k 10 200 python tools/microbench/seg_stamps.py $n 512 "1,2,4,5,10,11" sparse || exit 1
done > gpurun_out/r04/seg_phases_loops2.txt 2>&1
cat gpurun_out/r04/seg_phases_loops2.txt
