cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_mh.py -q -x -k "streaming_chain" > gpurun_out/big_tests.log 2>&1; echo "rc=$?" >> gpurun_out/big_tests.log
tail -4 gpurun_out/big_tests.log
for n in 256 200 128; do
for pp in 0 1; do
MCD_MH_PER_PHASE=$pp timeout -k 10 300 python bench.py --kind mh --n $n --steps 4000 --warmup 400 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=$n MCD_MH_PER_PHASE=$pp us per lock step %.2f  %.1f M steps/s' % (d['mh']['us_per_lockstep'], d['value']/1e6))
"
done
done
