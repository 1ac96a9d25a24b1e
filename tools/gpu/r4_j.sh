cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for rep in 1 2; do
for n in 64 128 192 256; do
for lw in 2 4; do
  MCD_LOADERS=$lw timeout -k 10 200 python bench.py --no-mh --no-cpu-baseline --steps 20000 --warmup 1000 --dim $n | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n $n loaders $lw', round(d['roofline']['kernel_us_per_launch'], 3), 'us')" | tee -a gpurun_out/r04/j_loaders.txt
done
done
done
for lw in 2 4; do
  MCD_LOADERS=$lw timeout -k 10 200 python bench.py --no-mh --no-cpu-baseline --steps 20000 --warmup 1000 --chains 256 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n 256 x 256 chains loaders $lw', round(d['roofline']['kernel_us_per_launch'], 3), 'us')" | tee -a gpurun_out/r04/j_loaders.txt
done
