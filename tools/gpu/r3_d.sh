# round 3: phase stamps of the streaming chain kernel with the incremental path; launch geometries of the headline sweep at 512
# chains (MCD_GEOM); the counters rocprofv3 offers for telling Infinity-Cache hits from HBM reads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 120 python tools/microbench/mhbig_stamps.py 256 512 8000 > gpurun_out/r03/d_mhbig_stamps.txt 2>&1
cat gpurun_out/r03/d_mhbig_stamps.txt
for g in "" 21 41 42; do
  for i in 1 2; do
  MCD_GEOM=$g timeout -k 10 120 python bench.py --steps 20000 --warmup 500 --no-cpu-baseline --no-mh 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('geom=[$g] value %.1f M  us_per_step %.2f  kernel %.2f us' % (d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_us_per_launch']))
"
  done
done > gpurun_out/r03/d_geom.txt 2>&1
cat gpurun_out/r03/d_geom.txt
rocprofv3 -L > gpurun_out/r03/d_counters_all.txt 2>&1
grep -i -E "mall|dram|hbm|EA0_RDREQ|EA0_RD_|FETCH_SIZE|TCC_MISS|TCC_HIT|TCC_REQ\b" gpurun_out/r03/d_counters_all.txt | cut -c1-200 | head -60 > gpurun_out/r03/d_counters.txt
cat gpurun_out/r03/d_counters.txt | head -40
