cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/d_seg_phases.txt
for cfg in "2012 64 1,2,4,5,10,11 sparse" "2012 64 1 sparse" "1024 512 1,2,4,5,10,11 sparse"; do
  timeout -k 10 200 python tools/microbench/seg_stamps.py $cfg >> gpurun_out/r04/d_seg_phases.txt 2>> gpurun_out/r04/d_seg_phases.err
done
cat gpurun_out/r04/d_seg_phases.txt; tail -3 gpurun_out/r04/d_seg_phases.err
timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -q -x -k "not 1007-512" 2>&1 | tail -3
rm -f gpurun_out/r04/d_sparse_mh.jsonl
for cfg in "1007 512 3000" "513 512 3000" "1007 64 2000"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_mh_large.py $1 $2 $3 sparse >> gpurun_out/r04/d_sparse_mh.jsonl 2>> gpurun_out/r04/d_sparse_mh.err
done
cut -c1-200 gpurun_out/r04/d_sparse_mh.jsonl
