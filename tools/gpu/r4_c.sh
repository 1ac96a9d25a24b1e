cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python tools/debug/sparse_seg_debug.py > gpurun_out/r04/c_debug.log 2>&1; echo rc=$?; grep -A3 "^[0-9]" gpurun_out/r04/c_debug.log | cut -c1-200
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py -q -x --durations=6 > gpurun_out/r04/c_sparse_tests.log 2>&1; echo "sparse tests rc=$?"; tail -12 gpurun_out/r04/c_sparse_tests.log
timeout -k 10 300 python bench.py --no-mh --no-cpu-baseline > gpurun_out/r04/c_bench.json 2> gpurun_out/r04/c_bench.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/c_bench.json') if l.startswith('{')][0]); print('headline', d['value'], d['roofline']['kernel_us_per_launch'])"
timeout -k 10 600 python bench.py --kind e2e > gpurun_out/r04/c_bench_e2e.json 2> gpurun_out/r04/c_bench_e2e.err; echo "e2e rc=$?"; tail -3 gpurun_out/r04/c_bench_e2e.err; cut -c1-1500 gpurun_out/r04/c_bench_e2e.json
