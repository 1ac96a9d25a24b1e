# round 4: the streaming kernel's threshold between "sparse" (columns of L^-1) and "dense" (in-kernel sweep) proposals
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for ss in 4 8 12 16 24 32 64; do
  for nl in 65 129; do
  MCD_MH_SPARSE_SLOTS=$ss timeout -k 10 250 python tools/bench_mh_large.py $nl 512 6000 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('sparse_slots=$ss', r['n_nodes'], 'nodes', '%.2f us per lock step' % r['us_per_lockstep'])" || exit 1
  done
done | tee gpurun_out/r04/sparse_slots.txt
