cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for n in 256 128; do
echo "== before, $n"; MHBSTAMPLIB=libmhbigstamp_old.so timeout -k 10 200 python tools/microbench/mhbig_stamps.py $n 512 4000 || exit 1
echo "== likelihood wave draws the next proposal, $n"; timeout -k 10 200 python tools/microbench/mhbig_stamps.py $n 512 4000 || exit 1
done > gpurun_out/r04/mhbig_phases_spec.txt 2>&1
cat gpurun_out/r04/mhbig_phases_spec.txt
