# round 4, last pass: the whole -m gpu suite, smoke(), the contract line (default and the driver's arguments)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04/gpu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r04/bench_default_last.json 2> gpurun_out/r04/bench_default_last.log || { tail gpurun_out/r04/bench_default_last.log; exit 1; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_last.json 2> gpurun_out/r04/bench_driver_last.log || exit 1
python - <<'PY'
import json
for f in ("gpurun_out/r04/bench_default_last.json", "gpurun_out/r04/bench_driver_last.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, d["metric"], round(d["value"] / 1e6, 2), "M", d["unit"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "cpu", d["cpu_baseline"]["value"],
          "mh", round(d["mh"]["us_per_lockstep"], 3), "cfg5", round(d["mh_config5_share"]["us_per_lockstep"], 3), "sparse", [round(x["us_per_lockstep"], 3) for x in d["mh_sparse"]])
PY
