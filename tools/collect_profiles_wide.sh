#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  tools/collect_profiles_wide.sh r01
# Multiply form of the log-density (k_wide.hip): bench.py at 8192 and 32768 chains under rocprofv3 --kernel-trace --stats,
# FETCH_SIZE / WRITE_SIZE of the 8192-chain case in separate --pmc passes, and tools/bench_forms.py (both forms side by
# side).  Summaries -> gpurun_out/<tag>/summary/<tag>_wide_*, copied to profiles/ by hand.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/summary
export TMPDIR=/tmp
cd /tmp
A="python3 $ROOT/bench.py --chains 8192 --steps 500 --warmup 50 --no-cpu-baseline"
B="python3 $ROOT/bench.py --n 1024 --chains 32768 --steps 60 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wide_trace_a -- $A > $OUT/wide_bench_a.json 2> $OUT/wide_trace_a.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wide_trace_b -- $B > $OUT/wide_bench_b.json 2> $OUT/wide_trace_b.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/wide_pmc_fetch -- $A > /dev/null 2> $OUT/wide_pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/wide_pmc_write -- $A > /dev/null 2> $OUT/wide_pmc_write.log
cd $ROOT
$A > $OUT/wide_bench_a_plain.json 2> /dev/null
$B > $OUT/wide_bench_b_plain.json 2> /dev/null
python3 tools/bench_forms.py --n 128 256 --batch 512 1024 2048 4096 8192 16384 32768 > $OUT/forms.jsonl 2> $OUT/forms.log
python3 tools/bench_forms.py --n 512 1024 --batch 512 2048 8192 32768 --iters 30 >> $OUT/forms.jsonl 2>> $OUT/forms.log
python3 tools/bench_forms.py --tree --n 255 1023 --batch 512 2048 8192 32768 --iters 30 >> $OUT/forms.jsonl 2>> $OUT/forms.log
python3 - <<PY
import csv, glob, json, os, statistics
base, tag = "$OUT", "$TAG"
out = os.path.join(base, "summary")
def find(sub, suffix):
    fs = glob.glob(os.path.join(base, sub, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None
res = {}
for key, sub, bench in (("n256_b8192", "wide_trace_a", "wide_bench_a.json"), ("n1024_b32768", "wide_trace_b", "wide_bench_b.json")):
    rows = [r for r in csv.DictReader(open(find(sub, "_kernel_trace.csv"))) if "k_wide" in r["Kernel_Name"]]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    line = [json.loads(l) for l in open(os.path.join(base, bench)) if l.startswith("{")][0]
    res[key] = {"kernel": rows[0]["Kernel_Name"], "dispatches": len(rows), "avg_ns": statistics.mean(dur), "median_ns": statistics.median(dur),
                "min_ns": min(dur), "max_ns": max(dur), "workgroup_size": int(rows[0]["Workgroup_Size_X"]), "grid_size": int(rows[0]["Grid_Size_X"]),
                "vgpr": int(rows[0]["VGPR_Count"]), "accum_vgpr": int(rows[0]["Accum_VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]),
                "lds_bytes": int(rows[0]["LDS_Block_Size"]), "scratch_bytes": int(rows[0]["Scratch_Size"]), "bench_line_under_rocprof": line,
                "bench_line": [json.loads(l) for l in open(os.path.join(base, bench.replace(".json", "_plain.json"))) if l.startswith("{")][0]}
    st = list(csv.DictReader(open(find(sub, "_kernel_stats.csv"))))
    with open(os.path.join(out, f"{tag}_wide_{key}_kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=st[0].keys()); w.writeheader(); w.writerows(st)
traffic = {}
for name, sub in (("FETCH_SIZE", "wide_pmc_fetch"), ("WRITE_SIZE", "wide_pmc_write")):
    f = find(sub, "_counter_collection.csv")
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_wide" in r.get("Kernel_Name", "") and r.get("Counter_Name") == name] if f else []
    if vals:
        traffic[name] = {"dispatches": len(vals), "mean_kib": statistics.mean(vals), "median_kib": statistics.median(vals)}
if traffic:
    fetch = traffic.get("FETCH_SIZE", {}).get("mean_kib", 0.0) * 1024.0
    write = traffic.get("WRITE_SIZE", {}).get("mean_kib", 0.0) * 1024.0
    traffic["per_launch_bytes_raw"] = fetch + write
    traffic["per_launch_bytes_corrected"] = 2.0 * fetch + write
    traffic["algorithmic_bytes"] = 8192 * 256 * 8 + 8192 * 8 + 256 * 257 // 2 * 8
    traffic["note"] = "N = 256, 8192 chains; rocprofv3 --pmc, separate passes; corrected = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)"
res["pmc_traffic_n256_b8192"] = traffic
json.dump(res, open(os.path.join(out, f"{tag}_wide_summary.json"), "w"), indent=1)
open(os.path.join(out, f"{tag}_wide_forms.jsonl"), "w").write("".join(l for l in open(os.path.join(base, "forms.jsonl")) if l.startswith("{")))
print(json.dumps({k: {kk: vv for kk, vv in v.items() if not kk.startswith("bench_line")} for k, v in res.items()}, indent=1))
PY
