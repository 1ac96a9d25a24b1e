# round 4: the streaming kernel's likelihood wave draws the next proposal (MhbSpec): its tests, then lock steps at 65 .. 257 nodes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "streaming or chain_kernel or big or posterior or twin or heated" > gpurun_out/r04/big_tests.log 2>&1 || { tail -30 gpurun_out/r04/big_tests.log; exit 1; }
tail -2 gpurun_out/r04/big_tests.log
for nl in 33 65 100 129; do timeout -k 10 250 python tools/bench_mh_large.py $nl 512 6000 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['n_nodes'], 'nodes x', r['chains'], 'chains', '%.2f us per lock step' % r['us_per_lockstep'], '|', r['path'][:60])" || exit 1; done
timeout -k 10 200 python bench.py --kind mh --steps 8000 --warmup 800 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench --kind mh', round(d['ms_per_step'] * 1e3, 3), 'us per lock step')"
