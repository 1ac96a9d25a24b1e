cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_sparse.py -q -m gpu -x > gpurun_out/r03/h_sparse.log 2>&1; tail -3 gpurun_out/r03/h_sparse.log
out=gpurun_out/r03/h_sparse_bench.jsonl; : > $out
for cfg in "256 512" "1024 512" "2011 512" "2011 4096" "4001 512" "8001 512"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --kind sparse --n $1 --chains $2 --steps 300 --warmup 30 2>/dev/null | tail -1 >> $out
done
python - <<'PY'
import json
for l in open('gpurun_out/r03/h_sparse_bench.jsonl'):
    d=json.loads(l); s=d['sparse']; print(s['n'], s['nnz'], s['chains'], round(s['kernel_us_per_launch'],1),'us', round(d['roofline']['achieved'],1),'GB/s', round(s['fp64_tflops'],2),'TF')
PY
# the densified route at the sizes it exists for
for n in 256 1024; do timeout -k 10 120 python bench.py --n $n --chains 512 --steps 2000 --warmup 100 --no-cpu-baseline --no-mh 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('dense n=$n', round(d['roofline']['kernel_us_per_launch'],2),'us', d['config']['form'])
"; done
