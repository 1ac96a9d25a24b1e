# round 4: the chain wave draws the next step's proposal while the other waves evaluate the one in flight
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests/test_gpu_mh.py tests/test_gpu_sparse.py -q -m gpu -x > gpurun_out/r04/spec_tests.log 2>&1 || { tail -30 gpurun_out/r04/spec_tests.log; exit 1; }
tail -2 gpurun_out/r04/spec_tests.log
for args in "--dim 1024 --chains 512 --swap-period 2" "--sparse --dim 1024 --chains 512" "--sparse --dim 2012 --chains 512" "--dim 598 --chains 512" "--sparse --dim 12 --chains 512"; do
    timeout -k 10 200 python bench.py --kind mh $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
done
timeout -k 10 200 python tools/microbench/seg_stamps.py 1024 512 > gpurun_out/r04/seg_phases_spec.txt 2>&1 && timeout -k 10 200 python tools/microbench/seg_stamps.py 2012 512 "1,2,4,5,10,11" sparse >> gpurun_out/r04/seg_phases_spec.txt 2>&1
cat gpurun_out/r04/seg_phases_spec.txt
