#!/bin/bash
# Run ON THE GPU BOX: bench.py over dimensions, kernels and batch sizes; one JSON line each -> gpurun_out/<tag>/bench_matrix.jsonl
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/bench_matrix.jsonl
for N in 64 128 256 512 1024; do
  for K in logpdf grad tree tree_grad; do
    for B in 512 8192; do
      timeout -k 10 180 python3 $ROOT/bench.py --n $N --kind $K --chains $B --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/bench_matrix.jsonl || exit 1
    done
  done
done
for K in prior posterior; do
  timeout -k 10 180 python3 $ROOT/bench.py --n 256 --kind $K --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/bench_matrix.jsonl || exit 1
done
python3 - <<PY
import json
rows = [json.loads(l) for l in open("$OUT/bench_matrix.jsonl")]
print("%-10s %6s %6s %10s %12s %8s %8s" % ("kind", "N", "B", "us/launch", "evals/s", "HBM %", "fp64 %"))
for r in rows:
    c, f = r["config"], r["roofline"]
    print("%-10s %6d %6d %10.2f %12.4g %8.2f %8.2f" % (c["kernel"], c["n"], c["chains_per_gpu"], f["kernel_us_per_launch"], r["value"], 100 * f["frac"], 100 * f["fp64_frac"]))
PY
