cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ms
timeout -k 10 800 python -m pytest tests/test_gpu_mh.py -q -x > gpurun_out/ms_tests.log 2>&1; echo "rc=$?" >> gpurun_out/ms_tests.log
tail -3 gpurun_out/ms_tests.log
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ms -- python3 $GRAFT_REPO_ROOT/bench.py --kind mh --steps 4000 --warmup 400 > $GRAFT_REPO_ROOT/gpurun_out/ms/bench.json 2>/dev/null
python3 - <<PY
import csv,glob,json
l=[x for x in open("$GRAFT_REPO_ROOT/gpurun_out/ms/bench.json") if x.startswith("{")][0]
print("us per lock step", json.loads(l)["mh"]["us_per_lockstep"])
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ms/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
