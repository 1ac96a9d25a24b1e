#!/bin/bash
# kernel statistics of one device leapfrog step (255-node and 1023-node trees, 512 chains)
mkdir -p gpurun_out/r03
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for n in 128 512; do
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r03/prof_lf$n" -- python3 "$ROOT/tools/bench_leapfrog.py" $n 512 200 > "$ROOT/gpurun_out/r03/prof_lf$n.log" 2>&1
f=$(find "$ROOT/gpurun_out/r03/prof_lf$n" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$ROOT/gpurun_out/r03/lf${n}_kernel_stats.csv"
grep "^{" "$ROOT/gpurun_out/r03/prof_lf$n.log" | cut -c1-200
python3 - "$ROOT/gpurun_out/r03/lf${n}_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.2f' % (float(r['AverageNs'])/1e3)).rjust(9), 'us', r['Percentage'])
PY
done
