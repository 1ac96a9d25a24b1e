# round 4, second GPU pass: the sparse driver's segments (k_mh_segment_sparse.hip), the one-launch sparse form (k_sparse_quad), the
# dense segment kernel after its chain wave moved to a shared header; then timings of the sparse paths.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py -q -x --durations=8 > gpurun_out/r04/b_sparse_tests.log 2>&1; echo "sparse tests rc=$?" | tee -a gpurun_out/r04/b_sparse_tests.log
tail -25 gpurun_out/r04/b_sparse_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_mh.py -q -x -k "incremental or large_tree or streaming or per_phase or workgroup" --durations=5 > gpurun_out/r04/b_mh_tests.log 2>&1; echo "mh tests rc=$?" | tee -a gpurun_out/r04/b_mh_tests.log
tail -12 gpurun_out/r04/b_mh_tests.log
for cfg in "256 512" "1024 512" "2011 512" "2011 4096" "8001 512" "2011 64"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --kind sparse --dim $1 --chains $2 --steps 300 --warmup 30 >> gpurun_out/r04/b_sparse_bench.jsonl 2>> gpurun_out/r04/b_sparse_bench.err
done
MCD_SPARSE_QUAD=0 timeout -k 10 200 python bench.py --kind sparse --dim 2011 --chains 512 --steps 300 --warmup 30 >> gpurun_out/r04/b_sparse_bench_rowform.jsonl 2>> gpurun_out/r04/b_sparse_bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r04/b_sparse_bench.jsonl", "gpurun_out/r04/b_sparse_bench_rowform.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], d["sparse"]["n"], d["sparse"]["chains"], round(d["sparse"]["kernel_us_per_launch"], 2), "us", round(d["roofline"]["frac"], 4))
PY
for cfg in "1007 512 3000" "513 512 3000" "200 512 3000" "1007 64 2000" "1007 1024 2000" "24 512 3000"; do
  set -- $cfg
  timeout -k 10 300 python tools/bench_mh_large.py $1 $2 $3 sparse >> gpurun_out/r04/b_sparse_mh.jsonl 2>> gpurun_out/r04/b_sparse_mh.err
done
MCD_MH_SEGMENTS=0 timeout -k 10 300 python tools/bench_mh_large.py 1007 512 1000 sparse >> gpurun_out/r04/b_sparse_mh_noseg.jsonl 2>> gpurun_out/r04/b_sparse_mh.err
cat gpurun_out/r04/b_sparse_mh.jsonl gpurun_out/r04/b_sparse_mh_noseg.jsonl | cut -c1-330
tail -3 gpurun_out/r04/b_sparse_mh.err gpurun_out/r04/b_sparse_bench.err
