# kernel durations and start-to-start intervals of bench.py under rocprofv3 --kernel-trace: tools/gpu/trace_gaps.sh "<bench args>" <tag>
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
OUT=$ROOT/gpurun_out/$2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline $1 > $OUT/bench.json 2> $OUT/trace.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + "/trace/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_split" in r["Kernel_Name"] or "k_logpdf" in r["Kernel_Name"] or "k_tree" in r["Kernel_Name"] or "k_wide" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-1000:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gap = [g for g in gap if g < 20000]
print(rows[0]["Kernel_Name"][:60], "n=%d dur mean %.0f median %.0f ns; gap (end->next start) mean %.0f median %.0f ns; LDS %s VGPR %s" % (len(rows), st.mean(dur), st.median(dur), st.mean(gap), st.median(gap), rows[0].get("LDS_Block_Size"), rows[0].get("VGPR_Count")))
PY
