#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<tag>/) into small committed summaries (gpurun_out/<tag>/summary/,
copied to profiles/ by hand):  <tag>_kernel_stats.csv, <tag>_kernel_trace_summary.json, <tag>_pmc_traffic.json.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced streaming reads, which is the access
shape of both the factor stream and (8 B/lane, uncalibrated) the chain vectors -- the corrected figure
doubles it and both numbers are kept.
"""
import csv
import glob
import json
import os
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", tag)
out = os.path.join(base, "summary")
os.makedirs(out, exist_ok=True)


def find(sub, suffix):
    fs = glob.glob(os.path.join(base, sub, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None


summary = {}
f = find("trace", "_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
f = find("trace", "_kernel_trace.csv")
if f:
    rows = [r for r in csv.DictReader(open(f)) if "mcd::" in r["Kernel_Name"]]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    summary = {"kernel": rows[0]["Kernel_Name"], "dispatches": len(rows), "avg_ns": statistics.mean(dur),
               "median_ns": statistics.median(dur), "min_ns": min(dur), "max_ns": max(dur),
               "workgroup_size": int(rows[0]["Workgroup_Size_X"]), "grid_size": int(rows[0]["Grid_Size_X"]),
               "vgpr": int(rows[0]["VGPR_Count"]), "accum_vgpr": int(rows[0]["Accum_VGPR_Count"]),
               "sgpr": int(rows[0]["SGPR_Count"]), "lds_bytes": int(rows[0]["LDS_Block_Size"]),
               "scratch_bytes": int(rows[0]["Scratch_Size"])}
    json.dump(summary, open(os.path.join(out, f"{tag}_kernel_trace_summary.json"), "w"), indent=1)

traffic = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = find(sub, "_counter_collection.csv")
    if not f:
        continue
    vals = []
    for r in csv.DictReader(open(f)):
        if "mcd::" in r.get("Kernel_Name", "") and r.get("Counter_Name") == name:
            vals.append(float(r["Counter_Value"]))
    if vals:
        traffic[name] = {"dispatches": len(vals), "mean_kib": statistics.mean(vals), "median_kib": statistics.median(vals),
                         "min_kib": min(vals), "max_kib": max(vals)}
if traffic:
    fetch = traffic.get("FETCH_SIZE", {}).get("mean_kib", 0.0) * 1024.0
    write = traffic.get("WRITE_SIZE", {}).get("mean_kib", 0.0) * 1024.0
    traffic["per_launch_bytes_raw"] = fetch + write
    traffic["per_launch_bytes_corrected"] = 2.0 * fetch + write   # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    traffic["note"] = ("rocprofv3 --pmc, separate passes; KiB per dispatch averaged over the bench's dispatches; corrected = "
                       "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)")
    json.dump(traffic, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
f = find("mh_trace", "_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, f"{tag}_mh_kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
for name in ("mh_steps.json", "bench_default.json"):
    src = os.path.join(base, name)
    if os.path.exists(src):
        lines = [l for l in open(src) if l.startswith("{")]
        open(os.path.join(out, f"{tag}_{name}"), "w").write("".join(lines))
print(json.dumps({"trace": summary, "traffic": traffic}, indent=1))
