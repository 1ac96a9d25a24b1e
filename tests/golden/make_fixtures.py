#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's own test INPUT data.

Run in the build container only (needs /root/reference):   python tests/golden/make_fixtures.py

For each dataset under /root/reference/tests/<NN>-leaves-*/ this restates `prepare`
(oracle/prepare.py, following app/Main.hs:159-307) on data/test.treelist and stores
  mu, sigma, sigma_inv, logdet              operands of the likelihood closure
  parent, names, mean_lengths               topology (pre-order) and the mean tree
  X, ll_X                                   retained posterior sample vectors + perturbed copies, and their
                                            log-likelihoods from the C oracle (Sigma^-1 form, Probability.hs:169)
  H, R, tH, rMu, ll_S, lj_S                 chain states (initWith + jitter), their
                                            log-likelihoods and log root-branch Jacobians
  gH, gR, gtH, grMu                         analytic gradients for the first 8 states
  prior_*, cal, con, brace_*, lp_<model>    the dataset's calibrations / constraints / braces on pre-order ids, drawn
                                            birth/death/variance/height parameters, and the log prior (+ its three
                                            blocks) of every chain state under each relaxed-clock model (prior_oracle.c)
The reference commits NO expected outputs for this path (parity unpinned); what pins these numbers
is (a) the C oracle, (b) the independent numpy twin and (c) scipy.stats.multivariate_normal, all of
which must agree here before a fixture is written.
"""
import os
import sys

import numpy as np
from scipy.stats import multivariate_normal

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from oracle import prepare as P  # noqa: E402

REF = "/root/reference/tests"
DATASETS = {
    "06-leaves-constant-rate": ("data/test.treelist", "data/time.tree"),
    "10-leaves-autocorrelated-rate": ("data/test.treelist", "data/time.alpha.tree"),
    "12-leaves-variable-rate": ("data/test.treelist", "data/time.alpha.rotated.tree"),
    "24-leaves-braces": ("data/test.treelist", "data/time.relabelled.tree"),
    "25-leaves-bastien": ("data/alignment.fasta.trees.only", "data/time.tree"),
}
N_SAMPLES = 48
N_STATES = {"06-leaves-constant-rate": 16, "10-leaves-autocorrelated-rate": 16,
            "12-leaves-variable-rate": 64, "24-leaves-braces": 128, "25-leaves-bastien": 32}


def jittered_states(prep, n_states, seed):
    """states initialised as initWith (app/Definitions.hs:96-123), then jittered:
    log-normal sigma 0.05 on rates, tH, rMu; every internal node keeps its initial height ratio to its
    parent times a log-normal factor (so parents stay above children).  Chain 0 is the un-jittered
    initial state except that rMu is set to the scale at which the distances match mu (the sampler's
    burn-in finds that scale; at rMu = 1 the log-likelihood is astronomically negative)."""
    rng = np.random.default_rng(seed)
    s0 = P.init_state(prep.parent, prep.mean_lengths)
    par = prep.parent
    nn = len(par)
    is_leaf = np.ones(nn, bool)
    is_leaf[par[1:]] = False
    h0 = s0["heights"]
    H = np.zeros((n_states, nn))
    H[:, 0] = 1.0
    for v in range(1, nn):
        if is_leaf[v]:
            continue
        frac = h0[v] / h0[par[v]]
        jit = np.exp(0.05 * rng.standard_normal(n_states))
        jit[0] = 1.0
        H[:, v] = H[:, par[v]] * np.clip(frac * jit, 0.02, 0.98)
    H[0] = h0
    R = np.exp(0.05 * rng.standard_normal((n_states, nn)))
    R[0, :] = 1.0
    R[:, 0] = 0.0
    d0 = P.distances_np(par, h0, s0["rates"], 1.0, 1.0)
    scale = float(prep.mu.sum() / d0.sum())
    tH = np.exp(0.05 * rng.standard_normal(n_states))
    tH[0] = 1.0
    rMu = scale * np.exp(0.05 * rng.standard_normal(n_states))
    rMu[0] = scale
    return H, R, tH, rMu


def mrca(parent, names, a, b):
    ia, ib = names.index(a), names.index(b)
    anc = set()
    v = ia
    while v >= 0:
        anc.add(v)
        v = int(parent[v])
    v = ib
    while v not in anc:
        v = int(parent[v])
    return v


def load_node_priors(dirname, parent, names):
    """calibrations.csv / constraints.csv / braces.json of a reference test directory -> tuples on pre-order ids."""
    import csv
    import json

    fnum = lambda x: float(x) if x.strip() else None
    cals, cons, brs = [], [], []
    p = os.path.join(dirname, "calibrations.csv")
    if os.path.exists(p):
        for r in list(csv.reader(open(p)))[1:]:
            if r:
                cals.append((mrca(parent, names, r[1], r[2]), fnum(r[3]), fnum(r[4]) or 0.0, fnum(r[5]), fnum(r[6]) or 0.0))
    p = os.path.join(dirname, "constraints.csv")
    if os.path.exists(p):
        for r in list(csv.reader(open(p)))[1:]:
            if r:
                cons.append((mrca(parent, names, r[1], r[2]), mrca(parent, names, r[3], r[4]), float(r[5])))
    p = os.path.join(dirname, "braces.json")
    if os.path.exists(p):
        for b in json.load(open(p)):
            brs.append(([mrca(parent, names, x, y) for x, y in b["braceDataNodes"]], float(b["braceDataStandardDeviation"])))
    return cals, cons, brs


MODELS = ["UncorrelatedGamma", "UncorrelatedLogNormal", "UncorrelatedWhiteNoise", "AutocorrelatedLogNormal"]


def prior_fixture(name, prep, H, R, rMu):
    """Log priors (app/Probability.hs:127-150) of the fixture's chain states under the dataset's own calibrations,
    constraints and braces, for every relaxed-clock model; the scalar parameters the likelihood does not use
    (birth/death rate, rate variance, absolute tree height) are drawn here."""
    d = os.path.join(REF, name, "data")
    cals, cons, brs = load_node_priors(d, prep.parent, list(prep.names))
    root = [c for c in cals if c[0] == 0]
    ht = (root[0][1] + root[0][3]) / 2.0 if (len(root) == 1 and root[0][1] is not None and root[0][3] is not None) else 1.0
    rng = np.random.default_rng(1000 + int(name[:2]))
    B = len(rMu)
    birth = np.exp(0.4 * rng.standard_normal(B))
    death = np.exp(0.4 * rng.standard_normal(B))
    rvar = np.exp(0.5 * rng.standard_normal(B)) * 0.3
    tH = ht * np.exp(0.1 * rng.standard_normal(B))
    death[1] = birth[1] * (1 + 1e-9)                      # near-critical branch of the birth-death prior
    out = dict(prior_ht=np.float64(ht), prior_birth=birth, prior_death=death, prior_rvar=rvar, prior_tH=tH,
               cal=np.array([[c[0], c[1] is not None, c[1] or 0.0, c[2], c[3] is not None, c[3] or 0.0, c[4]] for c in cals], float).reshape(-1, 7),
               con=np.array([[k[0], k[1], k[2]] for k in cons], float).reshape(-1, 3),
               brace_ptr=np.concatenate([[0], np.cumsum([len(b[0]) for b in brs])]).astype(np.int32),
               brace_nodes=np.array([n for b in brs for n in b[0]], np.int32), brace_sd=np.array([b[1] for b in brs], float))
    for m in MODELS:
        spec = O.PriorSpec(prep.parent, ht, m, cals, cons, brs)
        vals = np.array([O.prior(spec, birth[b], death[b], tH[b], H[b], rMu[b], rvar[b], R[b]) for b in range(B)], dtype=object)
        out["lp_" + m] = np.array([v[0] for v in vals])
        out["lpc_" + m] = np.stack([v[1] for v in vals])
    return out


def main():
    outdir = os.path.dirname(os.path.abspath(__file__))
    for name, (tl, rt) in DATASETS.items():
        prep = P.prepare(os.path.join(REF, name, tl), os.path.join(REF, name, rt))
        n = len(prep.mu)
        rng = np.random.default_rng(n)
        X = prep.samples[:N_SAMPLES].copy()
        Xp = X * np.exp(0.1 * rng.standard_normal(X.shape))
        X = np.concatenate([X, Xp, prep.mu[None, :]])
        ll_c = O.logpdf_full_batch(prep.mu, prep.sigma_inv, prep.logdet, X)
        ll_np = np.array([P.logpdf_full_np(prep.mu, prep.sigma_inv, prep.logdet, x) for x in X])
        ll_sp = multivariate_normal(mean=prep.mu, cov=prep.sigma).logpdf(X)
        L = O.cholesky(prep.sigma)
        ll_ch = O.logpdf_chol_batch(prep.mu, L, X)
        tol = 1e-10 * np.maximum(1.0, np.abs(ll_c))
        assert np.all(np.abs(ll_c - ll_np) <= tol), name
        assert np.all(np.abs(ll_c - ll_sp) <= 10 * tol), (name, np.max(np.abs(ll_c - ll_sp)))
        assert np.all(np.abs(ll_c - ll_ch) <= 10 * tol), (name, np.max(np.abs(ll_c - ll_ch)))

        H, R, tH, rMu = jittered_states(prep, N_STATES[name], seed=int(name[:2]))
        ll_S, lj_S = O.tree_loglik_full_batch(prep.parent, H, R, tH, rMu, prep.mu, prep.sigma_inv, prep.logdet)
        for b in range(len(tH)):
            d = P.distances_np(prep.parent, H[b], R[b], tH[b], rMu[b])
            ref = P.logpdf_full_np(prep.mu, prep.sigma_inv, prep.logdet, d)
            assert abs(ref - ll_S[b]) <= 1e-9 * max(1.0, abs(ref)), (name, b, ref, ll_S[b])
        ng = 8
        gH = np.zeros((ng, len(prep.parent))); gR = np.zeros_like(gH); gt = np.zeros(ng); gm = np.zeros(ng)
        for b in range(ng):
            gH[b], gR[b], gt[b], gm[b] = O.tree_grad_full(prep.parent, H[b], R[b], tH[b], rMu[b], prep.mu, prep.sigma_inv)
        pri = prior_fixture(name, prep, H, R, rMu)
        out = os.path.join(outdir, name + ".npz")
        np.savez_compressed(
            out, mu=prep.mu, sigma=prep.sigma, sigma_inv=prep.sigma_inv, logdet=np.float64(prep.logdet),
            parent=prep.parent, names=np.array(prep.names), mean_lengths=prep.mean_lengths,
            n_trees_kept=np.int64(len(prep.samples)), X=X, ll_X=ll_c, H=H, R=R, tH=tH, rMu=rMu, ll_S=ll_S, lj_S=lj_S,
            gH=gH, gR=gR, gtH=gt, grMu=gm, **pri)
        print(f"{name}: n={n} kept={len(prep.samples)} cond={np.linalg.cond(prep.sigma):.3g} logdet={prep.logdet:.6f} "
              f"ll(mu)={ll_c[-1]:.6f}  ll_S[0]={ll_S[0]:.4f} median ll_S={np.median(ll_S):.2f} -> {os.path.getsize(out)} B")


if __name__ == "__main__":
    main()
