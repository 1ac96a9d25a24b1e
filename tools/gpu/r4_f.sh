cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -q -x -k "not 1007-512" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py -q -x -k "prior_waves" 2>&1 | tail -3
rm -f gpurun_out/r04/f_seg_phases.txt
for cfg in "2012 64 1,2,4,5,10,11 sparse" "1024 512 1,2,4,5,10,11 sparse"; do
  timeout -k 10 200 python tools/microbench/seg_stamps.py $cfg >> gpurun_out/r04/f_seg_phases.txt 2>> gpurun_out/r04/f_seg_phases.err
done
cat gpurun_out/r04/f_seg_phases.txt; tail -3 gpurun_out/r04/f_seg_phases.err
for cfg in "1007 512 3000 sparse" "513 512 3000 sparse" "1007 64 2000 sparse" "200 512 3000 sparse" "24 512 3000 sparse" "7 128 3000 sparse"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_mh_large.py $1 $2 $3 $4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n_nodes'], d['chains'], d['form'], round(d['us_per_lockstep'], 2), 'us', d['path'][:40])"
done
