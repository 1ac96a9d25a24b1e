#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  tools/collect_profiles_r03.sh
# Round 3: (1) the default bench workload (N = 256, 512 chains): kernel trace + stats, FETCH_SIZE, WRITE_SIZE and the memory-side
# request split (all requests / DRAM-destined) in separate --pmc passes; (2) the lock-step Metropolis-Hastings run (257 nodes x 512
# chains: ONE k_mh_chain_big launch for the whole schedule): kernel trace + stats and the same counter passes -> traffic per lock
# step; (3) config 5's share of one GPU (1025 nodes x 512 chains) and the sparse form at N = 2011: kernel trace + stats.
# PMC is never combined with other trace domains (MI355X_MICROARCH.md).  Raw output under gpurun_out/r03/prof/; the condensed
# summaries land in gpurun_out/r03/prof/summary/ and are copied to profiles/ by hand.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r03/prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
N256="python3 $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-mh"
MH="python3 $ROOT/bench.py --kind mh --steps 8000 --warmup 800"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n256_trace -- $N256 > $OUT/n256_bench_trace.json 2> $OUT/n256_trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/n256_pmc_fetch -- $N256 > /dev/null 2> $OUT/n256_pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/n256_pmc_write -- $N256 > /dev/null 2> $OUT/n256_pmc_write.log
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/n256_pmc_dram -- $N256 > /dev/null 2> $OUT/n256_pmc_dram.log || true
echo "n256 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mh_trace -- $MH > $OUT/mh_bench_trace.json 2> $OUT/mh_trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/mh_pmc_fetch -- $MH > /dev/null 2> $OUT/mh_pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/mh_pmc_write -- $MH > /dev/null 2> $OUT/mh_pmc_write.log
MCD_MH_INCREMENTAL=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/mh0_pmc_fetch -- $MH > /dev/null 2> $OUT/mh0_pmc_fetch.log
echo "mh done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5_trace -- python3 $ROOT/bench.py --kind mh --dim 1024 --chains 512 --swap-period 2 --swap-steps 500 --steps 2000 --warmup 200 > $OUT/cfg5_bench_trace.json 2> $OUT/cfg5_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sparse_trace -- python3 $ROOT/bench.py --kind sparse --dim 2011 --chains 512 --steps 300 --warmup 30 > $OUT/sparse_bench_trace.json 2> $OUT/sparse_trace.log
CFG5="python3 $ROOT/bench.py --kind mh --dim 1024 --chains 512 --steps 4000 --warmup 400"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cfg5_pmc_fetch -- $CFG5 > /dev/null 2> $OUT/cfg5_pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cfg5_pmc_write -- $CFG5 > /dev/null 2> $OUT/cfg5_pmc_write.log
MCD_MH_SEGMENTS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cfg5ns_pmc_fetch -- $CFG5 > /dev/null 2> $OUT/cfg5ns_pmc_fetch.log
MCD_MH_SEGMENTS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cfg5ns_pmc_write -- $CFG5 > /dev/null 2> $OUT/cfg5ns_pmc_write.log
echo "traces done"
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.log
python3 bench.py --kind mh --steps 8000 --warmup 800 > $OUT/bench_mh.json 2> $OUT/bench_mh.log
python3 bench.py --kind mh --dim 1024 --chains 512 --swap-period 2 --steps 31826 --warmup 1000 > $OUT/bench_cfg5_1gpu.json 2> $OUT/bench_cfg5_1gpu.log
python3 bench.py --kind sparse --dim 2011 --chains 512 --steps 300 --warmup 30 > $OUT/bench_sparse.json 2> $OUT/bench_sparse.log
python3 - "$OUT" <<'PY'
import csv, glob, json, os, statistics, sys
base = sys.argv[1]
out = os.path.join(base, "summary")
os.makedirs(out, exist_ok=True)
def find(sub, suffix):
    fs = glob.glob(os.path.join(base, sub, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None
def dominant(tag, pick=None):
    f = find(tag + "_trace", "_kernel_trace.csv")
    if not f: return None
    rows = [r for r in csv.DictReader(open(f)) if "mcd::k_" in r["Kernel_Name"] and "poison" not in r["Kernel_Name"]]
    names = {}
    for r in rows: names.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    kn = pick(names) if pick else max(names, key=lambda k: len(names[k]))
    dur = names[kn]
    r0 = [r for r in rows if r["Kernel_Name"] == kn][0]
    return {"kernel": kn, "dispatches": len(dur), "avg_ns": statistics.mean(dur), "median_ns": statistics.median(dur), "min_ns": min(dur), "max_ns": max(dur),
            "workgroup_size": int(r0["Workgroup_Size_X"]), "grid_size": int(r0["Grid_Size_X"]), "vgpr": int(r0["VGPR_Count"]),
            "accum_vgpr": int(r0.get("Accum_VGPR_Count", 0) or 0), "sgpr": int(r0["SGPR_Count"]), "lds_bytes": int(r0["LDS_Block_Size"]), "scratch_bytes": int(r0["Scratch_Size"])}
def counter(sub, name, match):
    f = find(sub, "_counter_collection.csv")
    if not f: return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if match in r.get("Kernel_Name", "") and r.get("Counter_Name") == name]
    return vals
traces, traffic = {}, {}
for tag in ("n256", "mh", "cfg5", "sparse"):
    pick = (lambda names: max(names, key=lambda k: sum(names[k]))) if tag != "n256" else None
    d = dominant(tag, pick)
    if d: traces[tag] = d
    f = find(tag + "_trace", "_kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, f"r03_{tag}_kernel_stats.csv"), "w") as g:
            w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
# n256: per launch
fe, wr = counter("n256_pmc_fetch", "FETCH_SIZE", "k_logpdf"), counter("n256_pmc_write", "WRITE_SIZE", "k_logpdf")
if fe and wr:
    f_, w_ = statistics.mean(fe) * 1024, statistics.mean(wr) * 1024
    traffic["n256"] = {"FETCH_SIZE": {"dispatches": len(fe), "mean_kib": statistics.mean(fe)}, "WRITE_SIZE": {"dispatches": len(wr), "mean_kib": statistics.mean(wr)},
                       "per_launch_bytes_raw": f_ + w_, "per_launch_bytes_corrected": 2 * f_ + w_}
rq, rd = counter("n256_pmc_dram", "TCC_EA0_RDREQ_sum", "k_logpdf"), counter("n256_pmc_dram", "TCC_EA0_RDREQ_DRAM_sum", "k_logpdf")
if rq and rd:
    traffic["n256"]["memory_side_read_requests"] = {"all": statistics.mean(rq), "dram_destined": statistics.mean(rd),
        "note": "TCC_EA0_RDREQ counts requests that leave the L2 towards the fabric; _DRAM = destined for the memory controllers, in front of which the Infinity Cache sits: "
                "the counters cannot tell an Infinity-Cache hit from an HBM read"}
# mh: one launch = steps + warm-up launch; traffic per lock step of the timed launch (the largest FETCH value)
steps = 8000
for tag, key in (("mh", "mh_257x512"), ("mh0", "mh_257x512_full_sweeps")):
    fe = counter(tag + "_pmc_fetch", "FETCH_SIZE", "k_mh_chain_big")
    if fe:
        big = max(fe) * 1024
        traffic[key] = {"FETCH_SIZE_kib_of_the_timed_launch": max(fe), "lock_steps": steps, "fetch_bytes_per_lock_step_corrected": 2 * big / steps}
wr = counter("mh_pmc_write", "WRITE_SIZE", "k_mh_chain_big")
if wr: traffic["mh_257x512"]["write_bytes_per_lock_step"] = max(wr) * 1024 / steps
# config 5's share of one GPU (1025 nodes x 512 chains): every kernel of the run, per lock step -- with the segment kernel and with two launches per step
for tag, key in (("cfg5", "mh_1025x512_segments"), ("cfg5ns", "mh_1025x512_two_launches_per_step")):
    fe, wr = counter(tag + "_pmc_fetch", "FETCH_SIZE", "mcd::k_"), counter(tag + "_pmc_write", "WRITE_SIZE", "mcd::k_")
    if fe and wr:
        n = 4400
        traffic[key] = {"lock_steps_incl_warmup": n, "fetch_bytes_per_lock_step_corrected": 2 * sum(fe) * 1024 / n, "write_bytes_per_lock_step": sum(wr) * 1024 / n}
traffic["note"] = ("rocprofv3 --pmc, separate passes per counter; KiB per dispatch; corrected = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B "
                   "requests at 64 B).  mh_*: the whole schedule is ONE k_mh_chain_big launch; per lock step = counter of the timed launch / its lock steps")
json.dump(traffic, open(os.path.join(out, "r03_pmc_traffic.json"), "w"), indent=1)
json.dump(traces, open(os.path.join(out, "r03_kernel_trace_summary.json"), "w"), indent=1)
for name in ("bench_default.json", "bench_driver.json", "bench_mh.json", "bench_cfg5_1gpu.json", "bench_sparse.json"):
    src = os.path.join(base, name)
    if os.path.exists(src):
        open(os.path.join(out, "r03_" + name), "w").write("".join(l for l in open(src) if l.startswith("{")))
print(json.dumps({"traces": {k: (v["kernel"][:60], v["dispatches"], round(v["avg_ns"]), v["vgpr"], v["accum_vgpr"], v["scratch_bytes"], v["lds_bytes"]) for k, v in traces.items()},
                  "traffic": traffic}, indent=1)[:3000])
PY
