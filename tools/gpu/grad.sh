cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_split.py -q -x -k "gradient or config5" --durations=8 > gpurun_out/grad_tests.log 2>&1; echo "rc=$?" >> gpurun_out/grad_tests.log
tail -15 gpurun_out/grad_tests.log
for n in 1024 512 256; do
 for k in grad tree_grad; do
  for sp in 1 0; do
   echo "n=$n kind=$k MCD_SPLIT=$sp"
   MCD_SPLIT=$sp timeout -k 10 120 python bench.py --n $n --kind $k --steps 2000 --warmup 200 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   us/step', d['ms_per_step']*1e3, d['config'].get('form'))
"
  done
 done
done
