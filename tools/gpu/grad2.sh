cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 1024 512 256; do
  for k in grad tree_grad; do
   timeout -k 10 120 python bench.py --n $n --kind $k --steps 2000 --warmup 200 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=$n $k us/step %.2f' % (d['ms_per_step']*1e3), d['config'].get('form'))
"
  done
done
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/r2_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_gpu_tests.log
tail -20 gpurun_out/r2_gpu_tests.log
