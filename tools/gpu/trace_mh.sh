cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
OUT=$ROOT/gpurun_out/mhtrace
mkdir -p $OUT
MCD_MH_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_mh_large.py $1 $2 2000 > $OUT/bench.json 2> $OUT/trace.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics as st, collections
f = glob.glob(sys.argv[1] + "/trace/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"][:50]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items(): print("%-52s n=%6d mean %8.0f ns median %8.0f" % (k, len(v), st.mean(v), st.median(v)))
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows[-3000:], rows[-2999:])]
print("gap between consecutive kernels: mean %.0f median %.0f ns" % (st.mean(gaps), st.median(gaps)))
PY
cat $OUT/bench.json | tail -1
