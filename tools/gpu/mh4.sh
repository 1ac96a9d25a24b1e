cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_mh.py -q -x -k "beside or large_tree or per_phase" > gpurun_out/mh4_tests.log 2>&1; echo "rc=$?" >> gpurun_out/mh4_tests.log
tail -4 gpurun_out/mh4_tests.log
for pr in 1 0; do
MCD_MH_PRIOR=$pr timeout -k 10 300 python bench.py --kind mh --n 1024 --steps 2000 --warmup 200 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=1024 MCD_MH_PRIOR=$pr us per lock step %.2f  %.1f M steps/s' % (d['mh']['us_per_lockstep'], d['value']/1e6))
"
MCD_MH_PRIOR=$pr timeout -k 10 300 python bench.py --kind mh --n 512 --steps 2000 --warmup 200 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=512 MCD_MH_PRIOR=$pr us per lock step %.2f  %.1f M steps/s' % (d['mh']['us_per_lockstep'], d['value']/1e6))
"
done
