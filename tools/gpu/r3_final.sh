#!/bin/bash
# The README's round-3 figures in one GPU session: gpurun -- bash tools/gpu/r3_final.sh   (output: gpurun_out/r03/final.txt)
mkdir -p gpurun_out/r03
out=gpurun_out/r03/final.txt
: > $out
pick='import sys,json
r=json.loads(sys.stdin.read().strip().split("\n")[-1])
print({k: r[k] for k in ("value","unit","ms_per_step") if k in r}, {k: round(v.get("us_per_lockstep", v.get("kernel_us_per_launch", 0)), 3) for k, v in r.items() if isinstance(v, dict) and ("us_per_lockstep" in v or "kernel_us_per_launch" in v)})'
echo "== python bench.py" >> $out; python bench.py | python -c "$pick" >> $out
echo "== python bench.py --steps 20 --warmup 5 (the driver's arguments)" >> $out; python bench.py --steps 20 --warmup 5 | python -c "$pick" >> $out
echo "== python bench.py --kind mh --steps 8000 --warmup 800" >> $out; python bench.py --kind mh --steps 8000 --warmup 800 | python -c "$pick" >> $out
echo "== python bench.py --kind mh --dim 1024 --chains 512 --swap-period 2 --steps 31826 --warmup 1000 (config 5's share of one GPU)" >> $out
python bench.py --kind mh --dim 1024 --chains 512 --swap-period 2 --steps 31826 --warmup 1000 | python -c "$pick" >> $out
echo "== Metropolis-Hastings lock steps by tree size, 512 chains (tools/bench_mh_large.py leaves chains steps)" >> $out
for nl in 12 25 32 33 65 100 129 136 193 257 400 513; do python tools/bench_mh_large.py $nl 512 6000 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['n_nodes'], 'nodes', '%.2f us per lock step' % r['us_per_lockstep'], '|', r['path'][:70])" >> $out; done
echo "== 1025 nodes x 2048 / 4096 chains" >> $out
for B in 2048 4096; do python tools/bench_mh_large.py 513 $B 1500 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['chains'], 'chains', '%.2f us per lock step' % r['us_per_lockstep'])" >> $out; done
echo "== sparse likelihood: 2013 nodes x 512 chains" >> $out
python tools/bench_mh_large.py 1007 512 2000 sparse | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.2f us per lock step' % r['us_per_lockstep'])" >> $out
cat $out
