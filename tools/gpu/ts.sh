cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mh.py -q -x -k "tree or mh or fixture or parity or beside" > gpurun_out/ts_tests.log 2>&1; echo "rc=$?" >> gpurun_out/ts_tests.log
tail -4 gpurun_out/ts_tests.log
for n in 256 128 64; do
timeout -k 10 120 python bench.py --n $n --kind tree --steps 5000 --warmup 500 --no-cpu-baseline --no-mh 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=$n tree kernel us %.2f' % (d['roofline']['kernel_us_per_launch']))
"
done
timeout -k 10 300 python bench.py --kind mh --steps 4000 --warmup 400 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('mh us per lock step %.2f  %.1f M steps/s' % (d['mh']['us_per_lockstep'], d['value']/1e6))
"
