#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  tools/collect_profiles_r02.sh
# For the default bench workload (N = 256, 512 chains) and for BASELINE config 5's per-GPU share (N = 1024, 512 chains):
# (1) rocprofv3 kernel-trace + stats, (2) FETCH_SIZE and (3) WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: the TCC
# block cannot hold both; PMC is never combined with other trace domains).  Raw output under gpurun_out/r02/; the condensed
# summaries (written by the python at the end) are copied to profiles/ by hand.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "n256:" "n1024:--n 1024" "tree255:--kind tree" "tree1023:--n 1024 --kind tree"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  CMD="python3 $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-mh $args"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -- $CMD > $OUT/${tag}_bench_trace.json 2> $OUT/${tag}_trace.log
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_pmc_fetch -- $CMD > $OUT/${tag}_bench_fetch.json 2> $OUT/${tag}_pmc_fetch.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_pmc_write -- $CMD > $OUT/${tag}_bench_write.json 2> $OUT/${tag}_pmc_write.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mh_trace -- python3 $ROOT/bench.py --kind mh --steps 2000 --warmup 200 > $OUT/mh_bench.json 2> $OUT/mh_trace.log
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
python3 bench.py --n 1024 > $OUT/bench_n1024.json 2> $OUT/bench_n1024.log
python3 bench.py --kind mh --steps 4000 --warmup 400 > $OUT/bench_mh.json 2> $OUT/bench_mh.log
python3 - "$OUT" <<'PY'
import csv, glob, json, os, statistics, sys
base = sys.argv[1]
out = os.path.join(base, "summary")
os.makedirs(out, exist_ok=True)
def find(sub, suffix):
    fs = glob.glob(os.path.join(base, sub, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None
traffic, traces = {}, {}
for tag in ("n256", "n1024", "tree255", "tree1023"):
    f = find(tag + "_trace", "_kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, f"r02_{tag}_kernel_stats.csv"), "w") as g:
            w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    f = find(tag + "_trace", "_kernel_trace.csv")
    if f:
        rows = [r for r in csv.DictReader(open(f)) if "mcd::k_" in r["Kernel_Name"] and "poison" not in r["Kernel_Name"]]
        names = {}
        for r in rows: names.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        kn = max(names, key=lambda k: len(names[k]))
        dur = names[kn]
        r0 = [r for r in rows if r["Kernel_Name"] == kn][0]
        traces[tag] = {"kernel": kn, "dispatches": len(dur), "avg_ns": statistics.mean(dur), "median_ns": statistics.median(dur), "min_ns": min(dur),
                       "max_ns": max(dur), "workgroup_size": int(r0["Workgroup_Size_X"]), "grid_size": int(r0["Grid_Size_X"]), "vgpr": int(r0["VGPR_Count"]),
                       "sgpr": int(r0["SGPR_Count"]), "lds_bytes": int(r0["LDS_Block_Size"]), "scratch_bytes": int(r0["Scratch_Size"])}
    t = {}
    for name, sub in (("FETCH_SIZE", "_pmc_fetch"), ("WRITE_SIZE", "_pmc_write")):
        f = find(tag + sub, "_counter_collection.csv")
        if not f: continue
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "mcd::k_" in r.get("Kernel_Name", "") and "poison" not in r["Kernel_Name"] and r.get("Counter_Name") == name]
        if vals: t[name] = {"dispatches": len(vals), "mean_kib": statistics.mean(vals), "median_kib": statistics.median(vals)}
    if t:
        fetch = t.get("FETCH_SIZE", {}).get("mean_kib", 0.0) * 1024.0
        write = t.get("WRITE_SIZE", {}).get("mean_kib", 0.0) * 1024.0
        t["per_launch_bytes_raw"] = fetch + write
        t["per_launch_bytes_corrected"] = 2.0 * fetch + write
        traffic[tag] = t
traffic["note"] = ("rocprofv3 --pmc, separate passes per counter and workload; KiB per dispatch averaged over the bench's dispatches; corrected = "
                   "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B requests at 64 B)")
json.dump(traffic, open(os.path.join(out, "r02_pmc_traffic.json"), "w"), indent=1)
json.dump(traces, open(os.path.join(out, "r02_kernel_trace_summary.json"), "w"), indent=1)
f = find("mh_trace", "_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "r02_mh_kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
for name in ("bench_default.json", "bench_n1024.json", "bench_mh.json"):
    src = os.path.join(base, name)
    if os.path.exists(src):
        open(os.path.join(out, "r02_" + name), "w").write("".join(l for l in open(src) if l.startswith("{")))
print(json.dumps({"traces": traces, "traffic": {k: v.get("per_launch_bytes_corrected") for k, v in traffic.items() if isinstance(v, dict)}}, indent=1))
PY
