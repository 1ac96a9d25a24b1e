# round 4: MH tests, then the lock-step figures of every persistent kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests/test_gpu_mh.py tests/test_gpu_sparse.py -q -m gpu -x > gpurun_out/r04/mh_tests.log 2>&1 || { tail -30 gpurun_out/r04/mh_tests.log; exit 1; }
tail -2 gpurun_out/r04/mh_tests.log
for args in "--sparse --dim 12" "--sparse --dim 254" "--sparse --dim 1024" "--sparse --dim 2012" "--dim 30" "--dim 256" "--dim 598" "--dim 1024 --swap-period 2"; do
    timeout -k 10 200 python bench.py --kind mh $args --chains 512 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$args]', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')" || exit 1
done | tee gpurun_out/r04/mh_matrix.txt
