cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for k in "1,2,4,5,10,11" "1" "4" "10" "2,5,11"; do
timeout -k 10 200 python tools/microbench/seg_stamps.py 1024 512 $k || exit 1
done > gpurun_out/r04/seg_phases_dense.txt 2>&1
cat gpurun_out/r04/seg_phases_dense.txt
