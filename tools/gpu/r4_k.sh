cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for rep in 1 2; do
for lib in "" tools/microbench/libsetprio.so; do
  MCD_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 200 python bench.py --no-mh --no-cpu-baseline --steps 20000 --warmup 1000 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lib [$lib]', round(d['roofline']['kernel_us_per_launch'], 3), 'us')"
done
done
timeout -k 10 300 python tools/microbench/headline_phases.py 256 512 > gpurun_out/r04/headline_phases.txt 2> gpurun_out/r04/headline_phases.err; cat gpurun_out/r04/headline_phases.txt
