cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for n in 12 1024; do
for pd in 1 0; do
echo "== prior_draws=$pd, $n"; MCD_MH_PRIOR_DRAWS=$pd timeout -k 10 200 python tools/microbench/seg_stamps.py $n 512 "1,2,4,5,10,11" sparse || exit 1
done; done > gpurun_out/r04/seg_phases_prior_draws.txt 2>&1
cat gpurun_out/r04/seg_phases_prior_draws.txt
