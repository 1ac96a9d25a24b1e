cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "MCD_SPLIT_VEC=1 MCD_SPLIT_NOROT=0" "MCD_SPLIT_VEC=1 MCD_SPLIT_NOROT=1" "MCD_SPLIT_VEC=0 MCD_SPLIT_NOROT=0" "MCD_SPLIT_VEC=0 MCD_SPLIT_NOROT=1"; do
for args in "256 512" "1024 512"; do echo "== $v $args" | tee -a gpurun_out/r2_stamps3.log; env $v timeout -k 10 120 python tools/microbench/split_stamps.py $args 2>&1 | grep -v amdgpu.ids | head -4 | tee -a gpurun_out/r2_stamps3.log; done
for args in "--n 256 --chains 512" "--n 1024 --chains 512"; do
  env $v timeout -k 10 120 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$v $args', d['config']['form'], round(d['roofline']['kernel_us_per_launch'],2),'us', round(d['value']/1e6,1),'M/s')
    except Exception as e: print('$args', 'ERR', l[:300])
" | tee -a gpurun_out/r2_stamps3.log
done
done
