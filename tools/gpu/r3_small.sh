#!/bin/bash
# small trees (<= 64 nodes: k_mh_chain.hip): lock steps with and without the likelihood wave
mkdir -p gpurun_out/r03
for lw in 1 0; do
for cfg in "12 64" "12 512" "25 512" "32 512" "32 128"; do
  set -- $cfg
  echo -n "LW=$lw leaves=$1 chains=$2 "; MCD_MH_CHAIN_LW=$lw python tools/bench_mh_large.py $1 $2 20000 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['n_nodes'], '%.3f' % r['us_per_lockstep'])"
done; done > gpurun_out/r03/small_lw.txt 2>&1
cat gpurun_out/r03/small_lw.txt
