cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lf
export TMPDIR=/tmp
for cfg in "128 512" "512 512" "128 4096"; do
  set -- $cfg
  python tools/bench_leapfrog.py $1 $2 300
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lf/a -- python3 $GRAFT_REPO_ROOT/tools/bench_leapfrog.py 128 512 300 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/lf/a/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
