#!/bin/bash
# The README's round-4 Metropolis-Hastings figures in one GPU session: gpurun -- bash tools/gpu/r4_final.sh  (output: gpurun_out/r04/final.txt)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
out=gpurun_out/r04/final.txt
: > $out
echo "== Metropolis-Hastings lock steps by tree size, dense likelihood (tools/bench_mh_large.py leaves chains steps)" >> $out
for spec in "12 64" "25 512" "32 512" "33 512" "65 512" "100 512" "129 512" "150 512" "300 512" "400 512" "513 512" "513 2048"; do set -- $spec
  timeout -k 10 250 python tools/bench_mh_large.py $1 $2 6000 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['n_nodes'], 'nodes x', r['chains'], 'chains', '%.2f us per lock step' % r['us_per_lockstep'], '|', r['path'][:60])" >> $out || exit 1; done
echo "== sparse likelihood" >> $out
for spec in "7 128" "7 512" "24 512" "64 512" "200 512" "513 512" "1007 512" "1007 64"; do set -- $spec
  timeout -k 10 250 python tools/bench_mh_large.py $1 $2 4000 sparse | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['n_nodes'], 'nodes x', r['chains'], 'chains', '%.2f us per lock step' % r['us_per_lockstep'], '|', r['path'][:60])" >> $out || exit 1; done
cat $out
