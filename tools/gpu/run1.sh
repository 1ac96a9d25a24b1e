set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_split.py -x -q -k "not stress" > gpurun_out/r2_split_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_split_tests.log
tail -5 gpurun_out/r2_split_tests.log
for args in "--n 256 --chains 512" "--n 256 --chains 512 --form sweep" "--n 1024 --chains 512" "--n 512 --chains 512" "--n 256 --chains 512 --kind tree" "--n 1024 --chains 512 --kind tree" "--n 256 --chains 64" "--n 1024 --chains 64"; do
  timeout -k 10 120 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline $args 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$args', d['config']['form'], round(d['roofline']['kernel_us_per_launch'],2),'us', round(d['value']/1e6,1),'M/s')
    except Exception as e: print('$args', 'ERR', l[:300])
" | tee -a gpurun_out/r2_bench1.log
done
