#!/bin/bash
# Tuning sweep over workgroup geometries (compute waves, loader waves, chains per wave).
# usage: geom_sweep.sh <lib.so> <n> <batch> <geom> [<geom> ...]     (geom = cw,lw,bt or "default")
export MCD_LIB_PATH=$PWD/tools/microbench/$1
N=$2
B=$3
shift 3
for G in "$@"; do
  if [ "$G" = default ]; then unset MCD_GEOM; else export MCD_GEOM=$G; fi
  timeout 120 python bench.py --n $N --chains $B --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N', $N, 'B', $B, 'geom $G', '%.4g evals/s' % d['value'], '%.2f us' % d['roofline']['kernel_us_per_launch'], 'fp64 %.3f' % d['roofline']['fp64_frac'])"
done
