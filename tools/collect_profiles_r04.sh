#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  tools/collect_profiles_r04.sh
# Round 4: kernel trace + stats and memory-side traffic (FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes; PMC never combined with another
# trace domain, MI355X_MICROARCH.md) for
#   (1) the default bench workload (N = 256 x 512 chains: k_logpdf<4,1,2,2>),
#   (2) the sparse form in one launch (k_sparse_quad): N = 2011 and N = 256, 512 chains,
#   (3) lock-step Metropolis-Hastings: 257 nodes (k_mh_chain_big), config 5's share (1025 nodes: k_mh_segment + prior waves), and the sparse
#       driver at 1025 and 2013 nodes (k_mh_segment_sparse): traffic per lock step of the whole run,
# then the bench lines themselves.  Raw output under gpurun_out/r04/prof/; the condensed summaries land in gpurun_out/r04/prof/summary/ and are
# copied to profiles/ by hand.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04/prof
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
N256="python3 $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-mh"
SP2011="python3 $ROOT/bench.py --kind sparse --dim 2011 --chains 512 --steps 300 --warmup 30"
SP256="python3 $ROOT/bench.py --kind sparse --dim 256 --chains 512 --steps 300 --warmup 30"
MH257="python3 $ROOT/bench.py --kind mh --steps 8000 --warmup 800"
CFG5="python3 $ROOT/bench.py --kind mh --dim 1024 --chains 512 --steps 4000 --warmup 400"
SMH1025="python3 $ROOT/bench.py --kind mh --sparse --dim 1024 --chains 512 --steps 4000 --warmup 400"
SMH2013="python3 $ROOT/bench.py --kind mh --sparse --dim 2012 --chains 512 --steps 4000 --warmup 400"
run3() {   # tag, command: kernel trace + stats, FETCH_SIZE, WRITE_SIZE
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -- "$@" > $OUT/${tag}_bench_trace.json 2> $OUT/${tag}_trace.log
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_pmc_fetch -- "$@" > /dev/null 2> $OUT/${tag}_pmc_fetch.log
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_pmc_write -- "$@" > /dev/null 2> $OUT/${tag}_pmc_write.log
  echo "$tag done"
}
run3 n256 $N256
run3 sparse2011 $SP2011
run3 sparse256 $SP256
run3 mh257 $MH257
run3 cfg5 $CFG5
run3 smh1025 $SMH1025
run3 smh2013 $SMH2013
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.log
python3 bench.py --kind mh --steps 8000 --warmup 800 > $OUT/bench_mh.json 2> $OUT/bench_mh.log
python3 bench.py --kind mh --dim 1024 --chains 512 --swap-period 2 --steps 31826 --warmup 1000 > $OUT/bench_cfg5_1gpu.json 2> $OUT/bench_cfg5_1gpu.log
python3 bench.py --kind mh --sparse --dim 2012 --chains 512 --steps 8000 --warmup 800 > $OUT/bench_mh_sparse_2013.json 2> $OUT/bench_mh_sparse_2013.log
python3 bench.py --kind mh --sparse --dim 1024 --chains 512 --steps 8000 --warmup 800 > $OUT/bench_mh_sparse_1025.json 2> $OUT/bench_mh_sparse_1025.log
python3 bench.py --kind mh --dim 1024 --chains 512 --swap-period 2 --steps 31826 --warmup 1000 --tune-periods 20 > $OUT/bench_cfg5_1gpu_tuned.json 2> $OUT/bench_cfg5_1gpu_tuned.log
python3 bench.py --kind mh --sparse --dim 2012 --chains 512 --steps 8000 --warmup 800 --tune-periods 20 > $OUT/bench_mh_sparse_2013_tuned.json 2> $OUT/bench_mh_sparse_2013_tuned.log
python3 bench.py --kind mh --sparse --dim 1024 --chains 512 --steps 8000 --warmup 800 --tune-periods 20 > $OUT/bench_mh_sparse_1025_tuned.json 2> $OUT/bench_mh_sparse_1025_tuned.log
python3 bench.py --kind sparse --dim 2011 --chains 512 --steps 300 --warmup 30 > $OUT/bench_sparse.json 2> $OUT/bench_sparse.log
python3 bench.py --kind sparse --dim 256 --chains 512 --steps 300 --warmup 30 > $OUT/bench_sparse_256.json 2> $OUT/bench_sparse_256.log
python3 bench.py --kind e2e > $OUT/bench_e2e.json 2> $OUT/bench_e2e.log
echo "bench lines done"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, statistics, sys
base = sys.argv[1]
out = os.path.join(base, "summary")
os.makedirs(out, exist_ok=True)
def find(sub, suffix):
    fs = glob.glob(os.path.join(base, sub, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None
def kernels(tag):
    """every mcd:: kernel of the traced run: dispatches, total / mean duration, registers as the trace reports them"""
    f = find(tag + "_trace", "_kernel_trace.csv")
    if not f: return None
    rows = [r for r in csv.DictReader(open(f)) if "mcd::k_" in r["Kernel_Name"] and "poison" not in r["Kernel_Name"]]
    names = {}
    for r in rows: names.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
    res = {}
    for kn, lst in names.items():
        dur = [d for d, _ in lst]
        r0 = lst[0][1]
        res[kn] = {"dispatches": len(dur), "total_us": sum(dur) / 1e3, "avg_ns": statistics.mean(dur), "median_ns": statistics.median(dur), "min_ns": min(dur), "max_ns": max(dur),
                   "workgroup_size": int(r0["Workgroup_Size_X"]), "grid_size": int(r0["Grid_Size_X"]), "vgpr_count_trace": int(r0["VGPR_Count"]),
                   "accum_vgpr_count_trace": int(r0.get("Accum_VGPR_Count", 0) or 0), "sgpr_count_trace": int(r0["SGPR_Count"]),
                   "lds_bytes_static": int(r0["LDS_Block_Size"]), "scratch_bytes": int(r0["Scratch_Size"])}
    return res
def counter(sub, name, match="mcd::k_"):
    f = find(sub, "_counter_collection.csv")
    if not f: return None
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if match in r.get("Kernel_Name", "") and r.get("Counter_Name") == name]
def bench_line(name):
    try:
        return json.loads([l for l in open(os.path.join(base, name)) if l.startswith("{")][0])
    except Exception:
        return None
traces, traffic = {}, {}
for tag in ("n256", "sparse2011", "sparse256", "mh257", "cfg5", "smh1025", "smh2013"):
    k = kernels(tag)
    if k: traces[tag] = k
    f = find(tag + "_trace", "_kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, f"r04_{tag}_kernel_stats.csv"), "w") as g:
            w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
# dynamic LDS: the kernel trace shows the static group segment only; the library reports what its persistent kernels were launched with
for tag, name in (("mh257", "mh257_bench_trace.json"), ("cfg5", "cfg5_bench_trace.json"), ("smh1025", "smh1025_bench_trace.json"), ("smh2013", "smh2013_bench_trace.json")):
    d = bench_line(name)
    if d and tag in traces:
        traces[tag]["_dynamic_lds_bytes_of_the_persistent_kernel (mcd_mh_last_dynamic_lds)"] = d["mh"].get("lds_bytes_per_workgroup")
        traces[tag]["_path"] = d["mh"]["what"]
# per launch: n256 and the sparse form
for tag, key, match in (("n256", "n256", "k_logpdf"), ("sparse2011", "sparse_2011x512", "k_sparse_quad"), ("sparse256", "sparse_256x512", "k_sparse_quad")):
    fe, wr = counter(tag + "_pmc_fetch", "FETCH_SIZE", match), counter(tag + "_pmc_write", "WRITE_SIZE", match)
    if fe and wr:
        f_, w_ = statistics.mean(fe) * 1024, statistics.mean(wr) * 1024
        traffic[key] = {"kernel": match, "FETCH_SIZE": {"dispatches": len(fe), "mean_kib": statistics.mean(fe)}, "WRITE_SIZE": {"dispatches": len(wr), "mean_kib": statistics.mean(wr)},
                        "per_launch_bytes_raw": f_ + w_, "per_launch_bytes_corrected": 2 * f_ + w_}
# Metropolis-Hastings runs: every kernel of the run, per lock step (timed + warm-up steps)
for tag, key, n in (("mh257", "mh_257x512", 8800), ("cfg5", "mh_1025x512_segments", 4400), ("smh1025", "mh_sparse_1025x512", 4400), ("smh2013", "mh_sparse_2013x512", 4400)):
    fe, wr = counter(tag + "_pmc_fetch", "FETCH_SIZE"), counter(tag + "_pmc_write", "WRITE_SIZE")
    if fe and wr:
        traffic[key] = {"lock_steps_incl_warmup": n, "fetch_bytes_per_lock_step_corrected": 2 * sum(fe) * 1024 / n, "write_bytes_per_lock_step": sum(wr) * 1024 / n}
traffic["note"] = ("rocprofv3 --pmc, separate passes per counter; KiB per dispatch; corrected = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B "
                   "requests at 64 B).  mh_*: every mcd:: kernel of the run (set-up launches included), divided by its lock steps")
json.dump(traffic, open(os.path.join(out, "r04_pmc_traffic.json"), "w"), indent=1)
json.dump(traces, open(os.path.join(out, "r04_kernel_trace_summary.json"), "w"), indent=1)
for name in ("bench_default.json", "bench_driver.json", "bench_mh.json", "bench_cfg5_1gpu.json", "bench_mh_sparse_2013.json", "bench_mh_sparse_1025.json", "bench_sparse.json",
             "bench_sparse_256.json", "bench_e2e.json", "bench_cfg5_1gpu_tuned.json", "bench_mh_sparse_2013_tuned.json", "bench_mh_sparse_1025_tuned.json"):
    src = os.path.join(base, name)
    if os.path.exists(src):
        open(os.path.join(out, "r04_" + name), "w").write("".join(l for l in open(src) if l.startswith("{")))
brief = {}
for tag, k in traces.items():
    top = max((x for x in k if not x.startswith("_")), key=lambda x: k[x]["total_us"])
    brief[tag] = (top[:50], k[top]["dispatches"], round(k[top]["avg_ns"]), k[top]["vgpr_count_trace"], k[top]["scratch_bytes"], k.get("_dynamic_lds_bytes_of_the_persistent_kernel (mcd_mh_last_dynamic_lds)"))
print(json.dumps({"traces": brief, "traffic": traffic}, indent=1)[:4000])
PY
