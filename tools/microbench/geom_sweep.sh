#!/bin/bash
# Tuning sweep over workgroup geometries (compute waves, loader waves, chains per wave); needs libgeom.so.
export MCD_LIB_PATH=$PWD/tools/microbench/libgeom.so
for B in 512 131072; do
  for G in default 1,1,1 2,1,1 3,1,1 2,2,2 4,2,1 4,4,1 4,4,2 6,2,1 6,2,2 8,4,1 8,4,2; do
    if [ "$G" = default ]; then unset MCD_GEOM; else export MCD_GEOM=$G; fi
    timeout 120 python bench.py --chains $B --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $B, 'geom $G', '%.4g evals/s' % d['value'], '%.2f us' % d['roofline']['kernel_us_per_launch'], 'fp64 %.3f' % d['roofline']['fp64_frac'])"
  done
done
