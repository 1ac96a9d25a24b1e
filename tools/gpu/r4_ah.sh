# round 4: k_wide<CT, TREE> with three blocks per CU (no spill) against four (8 spilled registers): tree ll, multiply form, thousands of chains
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
true
true
for rep in 1 2; do
for lib in "" tools/microbench/libprev.so; do
  for args in "--kind tree --n 256 --chains 8192" "--kind tree --n 256 --chains 2048" "--kind tree --n 128 --chains 8192" "--kind tree --n 512 --chains 4096"; do
  MCD_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python bench.py $args --form multiply --steps 2000 --warmup 200 --no-cpu-baseline --no-mh 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lib [$lib] [$args]', round(d['ms_per_step'] * 1e3, 3), 'us per launch', round(d['roofline'].get('kernel_us_per_launch') or 0, 3), 'us kernel (HIP events)')" || exit 1
  done
done; done | tee gpurun_out/r04/wide_tree_ab.txt
