cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests/test_gpu_mh.py -q -x -k "streaming or chain_kernel" --durations=3 2>&1 | tail -6
for cfg in "129 512 8000 dense" "100 512 8000 dense" "65 512 8000 dense" "129 1024 4000 dense"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_mh_large.py $1 $2 $3 $4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n_nodes'], d['chains'], d['form'], round(d['us_per_lockstep'], 2), 'us', d['path'][:40])"
done
timeout -k 10 300 python bench.py --kind mh --steps 8000 --warmup 800 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('mh 257', d['mh']['us_per_lockstep'], d['value'])"
