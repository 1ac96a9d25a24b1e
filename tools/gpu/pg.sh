cd $GRAFT_REPO_ROOT
for cfg in "128 512" "128 1024" "128 2048" "128 4096"; do
  set -- $cfg
  python tools/bench_leapfrog.py $1 $2 300
done
