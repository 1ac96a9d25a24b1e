cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_mh.py -q -x -k "workgroup_per_chain or large_tree or beside or streaming" > gpurun_out/wg_tests.log 2>&1; echo "rc=$?" >> gpurun_out/wg_tests.log
tail -6 gpurun_out/wg_tests.log
for n in 1024 512 384; do
for wg in 1 0; do
MCD_MH_STEP_WG=$wg timeout -k 10 300 python bench.py --kind mh --n $n --steps 2000 --warmup 200 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=$n MCD_MH_STEP_WG=$wg us per lock step %.2f  %.1f M steps/s' % (d['mh']['us_per_lockstep'], d['value']/1e6))
"
done
done
