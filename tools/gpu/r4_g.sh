cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_mh.py -q -x -k "prior_waves or incremental or large_tree" 2>&1 | tail -3
rm -f gpurun_out/r04/g_seg_phases.txt
for cfg in "1024 512 1,2,4,5,10,11" "598 512 1,2,4,5,10,11"; do
  timeout -k 10 200 python tools/microbench/seg_stamps.py $cfg >> gpurun_out/r04/g_seg_phases.txt 2>> gpurun_out/r04/g_seg_phases.err
done
cat gpurun_out/r04/g_seg_phases.txt; tail -3 gpurun_out/r04/g_seg_phases.err
for cfg in "513 512 4000 dense" "400 512 4000 dense" "300 512 4000 dense" "150 512 4000 dense" "513 64 3000 dense" "513 2048 2000 dense"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_mh_large.py $1 $2 $3 $4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n_nodes'], d['chains'], d['form'], round(d['us_per_lockstep'], 2), 'us', d['path'][:40])"
done
timeout -k 10 300 python bench.py --kind mh --dim 1024 --chains 512 --swap-period 2 --steps 8000 --warmup 800 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg5 share', d['mh']['us_per_lockstep'], d['value'])"
timeout -k 10 300 python bench.py --kind mh --steps 8000 --warmup 800 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('mh 257', d['mh']['us_per_lockstep'], d['value'])"
