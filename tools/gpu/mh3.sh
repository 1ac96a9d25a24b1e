cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mh3
timeout -k 10 800 python -m pytest tests/test_gpu_mh.py -q -x --durations=4 > gpurun_out/mh3_tests.log 2>&1; echo "rc=$?" >> gpurun_out/mh3_tests.log
tail -9 gpurun_out/mh3_tests.log
for pr in 1 0; do
MCD_MH_PRIOR=$pr timeout -k 10 300 python bench.py --kind mh --steps 4000 --warmup 400 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('MCD_MH_PRIOR=$pr us per lock step %.2f  %.1f M steps/s' % (d['mh']['us_per_lockstep'], d['value']/1e6))
"
done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mh3 -- python3 $GRAFT_REPO_ROOT/bench.py --kind mh --steps 4000 --warmup 400 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/mh3/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
