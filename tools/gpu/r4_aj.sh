# round 4: the small-tree kernel's likelihood wave takes the clock block of the ln prior: tests, then lock steps at 23 .. 63 nodes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 200 python -u -m pytest tests/test_gpu_mh.py -q -m gpu -x -k "chain_kernel or lockstep" 2>&1 | tee gpurun_out/r04/aj_quick.log | tail -3; test ${PIPESTATUS[0]} -eq 0 || exit 1
timeout -k 10 800 python -u -m pytest tests/test_gpu_mh.py tests/test_gpu_prior.py tests/test_gpu_nuts.py -q -m gpu -x 2>&1 | tee gpurun_out/r04/aj_tests.log | tail -3; test ${PIPESTATUS[0]} -eq 0 || exit 1
for spec in "12 64" "12 512" "25 512" "32 512"; do set -- $spec
  timeout -k 10 250 python tools/bench_mh_large.py $1 $2 6000 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['n_nodes'], 'nodes x', r['chains'], 'chains', '%.2f us per lock step' % r['us_per_lockstep'], '|', r['path'][:60])" || exit 1; done
timeout -k 10 200 python bench.py --kind mh --dim 30 --chains 512 --tune-periods 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('31 nodes tuned', round(d['ms_per_step'] * 1e3, 3), 'us per lock step, acceptance', round(d['mh']['acceptance_rate'], 3))"
