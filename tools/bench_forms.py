"""Column sweep against the multiply form (k_wide.hip) of the log-density kernels: agreement and time per launch.

    python tools/bench_forms.py [--n 256 1024] [--batch 512 2048 8192 32768] [--tree]

Prints one JSON line per (N, batch): microseconds per launch of each form (device-resident inputs, hipGraph-free eager
launches timed with HIP events over `--iters` launches), fp64 FLOP rate of the multiply form against the vector peak,
and the largest difference between the two forms relative to |ll|.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mcmc_date_amd as M
from mcmc_date_amd import synthetic as S

FP64_PEAK = 78.6e12


def time_launches(fn, iters):
    import torch

    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    import torch

    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[256])
    ap.add_argument("--batch", type=int, nargs="+", default=[512, 2048, 8192, 32768])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--tree", action="store_true")
    ap.add_argument("--grad", action="store_true", help="time the gradient entry points (ll + gradient) instead of ll alone")
    ap.add_argument("--ct", type=int, default=0, help="informational: MCD_WIDE_CT must be set in the environment")
    a = ap.parse_args()
    L = M._capi.lib()
    dev = torch.device("cuda:0")
    for n in a.n:
        if a.tree:
            topo = S.random_topology((n + 3) // 2, seed=n)
            n = topo.n_nodes - 2
        mu, sigma = S.random_spd_problem(n, seed=n)
        lik = M.MvnLikelihood.from_covariance(mu, sigma)
        tl = lik.bind_tree(topo) if a.tree else None
        for B in a.batch:
            if a.tree and a.grad:
                st = S.random_states(topo, B, seed=B).to(dev)
                ll = torch.empty(B, dtype=torch.float64, device=dev)
                gH, gR, gt, gm = torch.empty_like(st.heights), torch.empty_like(st.rates), torch.empty_like(ll), torch.empty_like(ll)
                sp = M.likelihood._stream_ptr(0)

                def run():
                    M._capi.check(L.mcd_tree_grad_batch(tl._t, st.heights.data_ptr(), st.rates.data_ptr(), st.heights.stride(0), st.time_height.data_ptr(),
                                                        st.rate_mean.data_ptr(), B, 1, sp, ll.data_ptr(), gH.data_ptr(), gR.data_ptr(), gt.data_ptr(), gm.data_ptr()))
                    return gH
            elif a.tree:
                st = S.random_states(topo, B, seed=B).to(dev)
                run = lambda: tl.loglik(st, want_jacobian=True)[0]
            elif a.grad:
                X = torch.as_tensor(S.sample_chains(mu, sigma, B, seed=B), device=dev)
                ll = torch.empty(B, dtype=torch.float64, device=dev)
                G = torch.empty_like(X)
                sp = M.likelihood._stream_ptr(0)

                def run():
                    M._capi.check(L.mcd_mvn_grad_batch(lik._h, X.data_ptr(), X.stride(0), B, 1, sp, ll.data_ptr(), G.data_ptr(), G.stride(0)))
                    return G
            else:
                X = torch.as_tensor(S.sample_chains(mu, sigma, B, seed=B), device=dev)
                out = torch.empty(B, dtype=torch.float64, device=dev)
                run = lambda: (lik.logpdf_into(X, out), out)[1]
            res = {}
            vals = {}
            for name, form in (("sweep", 1), ("multiply", 2)):
                L.mcd_set_logpdf_form(form)
                vals[name] = run().cpu().numpy().copy()
                res[name + "_us"] = round(time_launches(run, a.iters), 2)
            L.mcd_set_logpdf_form(0)
            flops = n * (n + 1.0) * B * (2 if a.grad else 1)
            scale = np.abs(vals["sweep"]).max() if a.grad else np.maximum(1.0, np.abs(vals["sweep"]))
            diff = float(np.max(np.abs(vals["sweep"] - vals["multiply"]) / scale))
            print(json.dumps({"n": n, "batch": B, "tree": a.tree, "grad": a.grad, **res, "multiply_frac_fp64_peak": round(flops / (res["multiply_us"] * 1e-6) / FP64_PEAK, 4),
                              "sweep_frac_fp64_peak": round(flops / (res["sweep_us"] * 1e-6) / FP64_PEAK, 4), "max_rel_diff": diff,
                              "ct": os.environ.get("MCD_WIDE_CT", "auto")}), flush=True)


if __name__ == "__main__":
    main()
