#!/bin/bash
# kernel statistics of the segment path (1025 nodes x 512 chains): which launches a lock step is made of
mkdir -p gpurun_out/r03
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r03/prof_seg" -- python3 "$ROOT/tools/bench_mh_large.py" 513 512 3000 > "$ROOT/gpurun_out/r03/prof_seg.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/r03/prof_seg -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r03/r_seg_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03/r_seg_kernel_stats.csv')))
for r in rows[:10]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(7), ('%.2f' % (float(r['AverageNs'])/1e3)).rjust(9), 'us', r['Percentage'])
PY
