cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_mh.py -q -x -k "prior_waves or incremental or large_tree" --durations=5 > gpurun_out/r04/e_mh_tests.log 2>&1; echo "mh tests rc=$?"; tail -12 gpurun_out/r04/e_mh_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -q -x -k "not 1007-512" 2>&1 | tail -3
rm -f gpurun_out/r04/e_seg_phases.txt gpurun_out/r04/e_mh.jsonl
for cfg in "2012 64 1,2,4,5,10,11 sparse" "1024 512 1,2,4,5,10,11 sparse" "1024 512 1,2,4,5,10,11"; do
  timeout -k 10 200 python tools/microbench/seg_stamps.py $cfg >> gpurun_out/r04/e_seg_phases.txt 2>> gpurun_out/r04/e_seg_phases.err
done
cat gpurun_out/r04/e_seg_phases.txt
for w in 1 0; do
for cfg in "1007 512 3000 sparse" "513 512 3000 sparse" "1007 64 2000 sparse" "513 512 3000 dense" "300 512 3000 dense" "150 512 3000 dense"; do
  set -- $cfg
  MCD_MH_PRIOR_WAVES=$w timeout -k 10 300 python tools/bench_mh_large.py $1 $2 $3 $4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('prior waves $w', d['n_nodes'], d['chains'], d['form'], round(d['us_per_lockstep'], 2), 'us', d['path'][:40])" | tee -a gpurun_out/r04/e_mh.jsonl
done
done
