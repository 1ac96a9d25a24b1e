"""bench.py started the way the driver starts it -- and the multi-rank path of BASELINE.json config 5 run to its end.

`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) launches its own two ranks as fresh child processes; on a box with one
device they run as a rehearsal (both ranks on cuda:0, control collectives and the swap all-gather through gloo: RCCL refuses
two ranks on one device) and the line says so.  What is asserted: ONE parsed JSON line with n_gpus == 2, finite values, every
rank's own ms_per_step, and for the Metropolis-Hastings kind the invariant of the MC3 swap phase (the temperature ranks of
every group of four chains are a permutation -- bench.py asserts it on every rank before printing; a failed assertion is a
non-zero exit).  Reference analogue: `mc3 (MC3Settings (NChains 4) (SwapPeriod 2) (NSwaps 3))`, app/Main.hs:476-478.
The lines are kept under gpurun_out/r04/ (copied to profiles/r04_bench_rehearsal_*.json by hand)."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return p


def _one_line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert p.returncode == 0, f"bench.py exited with {p.returncode}\n{p.stderr[-3000:]}"
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}\n{p.stdout[-2000:]}"
    return json.loads(lines[0])


def _keep(name, line):
    d = os.path.join(ROOT, "gpurun_out", "r04")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass


def test_bare_multi_rank_start_without_a_gpu_fails_loudly():
    """No device: the launcher says so and exits non-zero (the product has no CPU path); with a device this is covered below."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a device is present: the GPU tests below run the launcher for real")
    p = _run_bench(["--gpus", "2", "--steps", "5", "--warmup", "1"], timeout=600)
    assert p.returncode != 0
    assert "needs a GPU" in (p.stderr + p.stdout)
    assert "torch.distributed.run" not in (p.stderr + p.stdout)       # the round-3 refusal is gone


@pytest.mark.gpu
def test_two_ranks_metropolis_hastings_with_the_mc3_swap_phase(gpu):
    """Config 5's command at rehearsal size: chains sharded over two ranks, swap phase every 200 lock steps."""
    p = _run_bench(["--gpus", "2", "--kind", "mh", "--dim", "256", "--chains", "64", "--swap-steps", "200", "--steps", "600", "--warmup", "100"])
    line = _one_line(p)
    _keep("bench_rehearsal_mh.json", line)
    assert line["n_gpus"] == 2 and line["steps"] == 600 and line["warmup"] == 100
    assert math.isfinite(line["value"]) and line["value"] > 0 and math.isfinite(line["ms_per_step"])
    r = line["ranks"]
    assert r["world_size"] == 2 and len(r["ms_per_step_per_rank"]) == 2 and all(math.isfinite(x) and x > 0 for x in r["ms_per_step_per_rank"])
    assert abs(max(r["ms_per_step_per_rank"]) - line["ms_per_step"]) <= 1e-9 * line["ms_per_step"]      # value = the MAX over the ranks
    assert r["launched_by"].startswith("bench.py")
    import torch
    if torch.cuda.device_count() < 2:
        assert r["rehearsal"] is True and r["devices"] == 1
    mc3 = line["mh"]["mc3"]
    assert mc3["ranks"] == 2 and mc3["phases_timed"] == 3 and mc3["period_lock_steps"] == 200
    assert mc3["bytes_gathered_per_phase"] == 2 * 3 * 64 * 8
    assert sum(mc3["swaps_tried"]) == (mc3["phases_timed"] + 1) * 3 * (2 * 64 // 4)       # the phase of the warm-up + the timed ones, 3 swaps x 32 groups
    assert 0 < sum(mc3["swaps_accepted"]) <= sum(mc3["swaps_tried"])


@pytest.mark.gpu
def test_two_ranks_default_kind(gpu):
    """The default line (--kind logpdf, N = 256 x 512 chains per rank) over two self-launched ranks."""
    p = _run_bench(["--gpus", "2", "--steps", "200", "--warmup", "50"])
    line = _one_line(p)
    _keep("bench_rehearsal_logpdf.json", line)
    assert line["n_gpus"] == 2 and line["metric"].startswith("MVN log-likelihood evals/sec")
    assert math.isfinite(line["value"]) and line["value"] > 0
    assert line["config"]["chains_per_gpu"] == 512 and line["config"]["n"] == 256
    assert len(line["ranks"]["ms_per_step_per_rank"]) == 2
    assert line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_one_rank_line_is_unchanged_in_shape(gpu):
    """N = 1 started bare: no launcher, no `ranks` field, the fields the contract names."""
    p = _run_bench(["--steps", "20", "--warmup", "5", "--no-mh", "--no-cpu-baseline"])
    line = _one_line(p)
    assert line["n_gpus"] == 1 and "ranks" not in line
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line


@pytest.mark.gpu
def test_tuned_metropolis_hastings_line(gpu):
    """`--kind mh --tune-periods T`: T auto-tuning periods before the clock (the reference samples with tuned proposals); the line says
    which acceptance rate the timed steps had, and tuning moves it towards `mcmc`'s targets (0.23 .. 0.44) from the initial parameters' rate."""
    rates = []
    for periods in ("0", "12"):
        line = _one_line(_run_bench(["--kind", "mh", "--sparse", "--dim", "60", "--chains", "64", "--steps", "1500", "--warmup", "100", "--tune-periods", periods,
                                     "--no-cpu-baseline"]))
        mh = line["mh"]
        assert mh["tune_periods"] == int(periods) and 0.0 < mh["acceptance_rate"] < 1.0 and math.isfinite(line["value"]) and line["value"] > 0
        assert "segments over a sparse precision matrix" in mh["what"]
        rates.append(mh["acceptance_rate"])
    assert 0.2 < rates[1] < 0.55 and abs(rates[1] - 0.35) < abs(rates[0] - 0.35) + 0.02, rates
