"""CPU oracle for the MVN log-likelihood path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/mvn_oracle.c header): the reference has no golden vectors for
this path and cannot be built here (Haskell, no GHC).  The oracle is a restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (mcmc-date_amd/) never does.

`lib()` returns the ctypes handle of oracle/liboracle.so (built by oracle/Makefile, or by
__graft_entry__.build()).  The thin wrappers below take/return numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBS = {}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class _OrmModel(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("parent", _ip), ("mu", _dp), ("sigma_inv", _dp), ("logdet", C.c_double),
                ("ht", C.c_double), ("clock_model", C.c_int32), ("ncal", C.c_int32), ("cal_node", _ip), ("cal_has_lo", _ip),
                ("cal_lo", _dp), ("cal_lo_p", _dp), ("cal_has_hi", _ip), ("cal_hi", _dp), ("cal_hi_p", _dp),
                ("ncon", C.c_int32), ("con_young", _ip), ("con_old", _ip), ("con_p", _dp),
                ("nbr", C.c_int32), ("br_ptr", _ip), ("br_nodes", _ip), ("br_sd", _dp),
                ("n_prop", C.c_int32), ("kind", _ip), ("node", _ip), ("n1", _ip), ("n2", _ip), ("jac_root", _ip), ("dim", _ip),
                ("p0", _dp), ("p1", _dp)]


def build(native: bool = False, force: bool = False) -> str:
    target = "liboracle_native.so" if native else "liboracle.so"
    subprocess.run(["make", "-s"] + (["-B"] if force else []) + ["-C", _HERE, target], check=True)
    _LIBS.pop(bool(native), None)
    return os.path.join(_HERE, target)


def lib(native: bool = False):
    key = bool(native)
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(_HERE, "liboracle_native.so" if native else "liboracle.so")
    src = os.path.join(_HERE, "mvn_oracle.c")
    src2 = os.path.join(_HERE, "prior_oracle.c")
    src3 = os.path.join(_HERE, "mh_oracle.c")
    newest = max(os.path.getmtime(p) for p in (src, src2, src3) if os.path.exists(p))
    if not os.path.exists(path) or newest > os.path.getmtime(path):
        build(native)
    # mh_oracle.c runs chains under OpenMP; a GPU box shows every host core but grants a share of about 16
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
    L = C.CDLL(path)
    L.orc_logpdf_full.restype = C.c_double
    L.orc_logpdf_full.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp]
    L.orc_logpdf_full_ld.restype = C.c_longdouble
    L.orc_logpdf_full_ld.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp]
    L.orc_quadform_full.restype = C.c_double
    L.orc_quadform_full.argtypes = [C.c_int, _dp, _dp, _dp]
    L.orc_logpdf_univariate.restype = C.c_double
    L.orc_logpdf_univariate.argtypes = [C.c_int, _dp, _dp, _dp]
    L.orc_logpdf_sparse.restype = C.c_double
    L.orc_logpdf_sparse.argtypes = [C.c_int, _dp, C.c_int64, _ip, _ip, _dp, C.c_double, _dp]
    L.orc_cholesky.restype = C.c_int
    L.orc_cholesky.argtypes = [C.c_int, _dp, _dp]
    L.orc_quadform_chol.restype = C.c_double
    L.orc_quadform_chol.argtypes = [C.c_int, _dp, _dp, _dp]
    L.orc_logpdf_chol.restype = C.c_double
    L.orc_logpdf_chol.argtypes = [C.c_int, _dp, _dp, _dp]
    L.orc_grad_full.restype = None
    L.orc_grad_full.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
    L.orc_height_to_length.restype = None
    L.orc_height_to_length.argtypes = [C.c_int, _ip, _dp, _dp]
    L.orc_get_branches.restype = C.c_int
    L.orc_get_branches.argtypes = [C.c_int, _ip, _dp, _dp]
    L.orc_sum_first_two.restype = None
    L.orc_sum_first_two.argtypes = [C.c_int, _dp, _dp]
    L.orc_distances.restype = C.c_int
    L.orc_distances.argtypes = [C.c_int, _ip, _dp, _dp, C.c_double, C.c_double, _dp]
    L.orc_tree_loglik_full.restype = C.c_int
    L.orc_tree_loglik_full.argtypes = [C.c_int, _ip, _dp, _dp, C.c_double, C.c_double, _dp, _dp, C.c_double, _dp]
    L.orc_log_jacobian_root_branch.restype = C.c_int
    L.orc_log_jacobian_root_branch.argtypes = [C.c_int, _ip, _dp, _dp, C.c_double, C.c_double, _dp]
    L.orc_tree_grad_full.restype = C.c_int
    L.orc_tree_grad_full.argtypes = [C.c_int, _ip, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_logpdf_full_batch.restype = None
    L.orc_logpdf_full_batch.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, C.c_int64, C.c_int64, _dp]
    L.orc_logpdf_full_batch_mt.restype = None
    L.orc_logpdf_full_batch_mt.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, C.c_int64, C.c_int64, _dp]
    L.orc_logpdf_chol_batch.restype = None
    L.orc_logpdf_chol_batch.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int64, C.c_int64, _dp]
    L.orc_grad_full_batch.restype = None
    L.orc_grad_full_batch.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int64, C.c_int64, _dp, C.c_int64]
    L.orc_tree_loglik_full_batch.restype = C.c_int
    L.orc_tree_loglik_full_batch.argtypes = [C.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int64, _dp, _dp]
    # prior_oracle.c
    L.orp_ln_exponential.restype = C.c_double; L.orp_ln_exponential.argtypes = [C.c_double, C.c_double]
    L.orp_ln_gamma.restype = C.c_double; L.orp_ln_gamma.argtypes = [C.c_double] * 3
    L.orp_ln_normal.restype = C.c_double; L.orp_ln_normal.argtypes = [C.c_double] * 3
    L.orp_calibrate_soft.restype = C.c_double
    L.orp_calibrate_soft.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double]
    L.orp_constrain_soft.restype = C.c_double; L.orp_constrain_soft.argtypes = [C.c_double] * 3
    L.orp_brace_soft.restype = C.c_double; L.orp_brace_soft.argtypes = [C.c_double, C.c_int, _dp]
    L.orp_compute_de.restype = None; L.orp_compute_de.argtypes = [C.c_double] * 5 + [_dp, _dp]
    L.orp_compute_de_near_critical.restype = None; L.orp_compute_de_near_critical.argtypes = [C.c_double] * 5 + [_dp, _dp]
    L.orp_birth_death.restype = C.c_double
    L.orp_birth_death.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _ip, _dp]
    L.orp_ln_lognormal_prime.restype = C.c_double; L.orp_ln_lognormal_prime.argtypes = [C.c_double] * 3
    L.orp_relaxed_clock.restype = C.c_double
    L.orp_relaxed_clock.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp]
    L.orp_prior.restype = C.c_double
    L.orp_prior.argtypes = ([C.c_double, C.c_int, C.c_int, _ip, C.c_double, C.c_double, C.c_double, _dp, C.c_double, C.c_double, _dp]
                            + [C.c_int, _ip, _ip, _dp, _dp, _ip, _dp, _dp] + [C.c_int, _ip, _ip, _dp] + [C.c_int, _ip, _ip, _dp] + [_dp])
    # mh_oracle.c
    _u32p = C.POINTER(C.c_uint32)
    L.orm_philox4x32.restype = None; L.orm_philox4x32.argtypes = [_u32p, _u32p, _u32p]
    L.orm_uniform_pair.restype = None; L.orm_uniform_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, _dp]
    L.orm_erfinv.restype = C.c_double; L.orm_erfinv.argtypes = [C.c_double]
    L.orm_tn_logpdf.restype = C.c_double; L.orm_tn_logpdf.argtypes = [C.c_double] * 5
    L.orm_tn_quantile.restype = C.c_double; L.orm_tn_quantile.argtypes = [C.c_double] * 5
    L.orm_gamma_draw.restype = C.c_double; L.orm_gamma_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_double, C.c_double]
    _mp = C.POINTER(_OrmModel)
    L.orm_run.restype = C.c_int
    L.orm_run.argtypes = ([_mp, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int64, _ip, C.c_int64, C.c_int32, C.c_uint64,
                           C.c_uint64, C.c_int64, _dp, _dp, _ip, _ip, _dp, _dp, _dp, _dp, C.POINTER(C.c_int8)])
    L.orm_tune.restype = None; L.orm_tune.argtypes = [_mp, C.c_int64, _dp, _ip, _ip]
    L.orm_propose_once.restype = C.c_int
    L.orm_propose_once.argtypes = [_mp, C.c_int, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    _LIBS[key] = L
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc == -2:
        raise OracleError("getBranches: Root node is not bifurcating.")  # app/Tools.hs:43
    if rc != 0:
        raise OracleError(f"oracle error {rc}")


# ---- raw-x entry points ---------------------------------------------------------------
def logpdf_full(mu, sigma_inv, logdet, x) -> float:
    mu, pm = _d(mu); P, pp = _d(sigma_inv); x, px = _d(x)
    return float(lib().orc_logpdf_full(len(mu), pm, pp, float(logdet), px))


def logpdf_full_ld(mu, sigma_inv, logdet, x) -> float:
    mu, pm = _d(mu); P, pp = _d(sigma_inv); x, px = _d(x)
    return float(lib().orc_logpdf_full_ld(len(mu), pm, pp, float(logdet), px))


def logpdf_full_batch(mu, sigma_inv, logdet, X, native=False, all_cores=False) -> np.ndarray:
    """X: [batch, n] chain-major.  all_cores: chains split over the host's cores (OpenMP; OMP_NUM_THREADS)."""
    mu, pm = _d(mu); P, pp = _d(sigma_inv); X, px = _d(X)
    out = np.empty(X.shape[0]); po = out.ctypes.data_as(_dp)
    f = lib(native).orc_logpdf_full_batch_mt if all_cores else lib(native).orc_logpdf_full_batch
    f(len(mu), pm, pp, float(logdet), px, X.shape[1], X.shape[0], po)
    return out


def cholesky(sigma) -> np.ndarray:
    S, ps = _d(sigma)
    n = S.shape[0]
    L = np.zeros((n, n)); pl = L.ctypes.data_as(_dp)
    rc = lib().orc_cholesky(n, ps, pl)
    if rc != 0:
        raise OracleError("covariance matrix is not positive definite")
    return L


def logpdf_chol_batch(mu, L, X) -> np.ndarray:
    mu, pm = _d(mu); L, pl = _d(L); X, px = _d(X)
    out = np.empty(X.shape[0]); po = out.ctypes.data_as(_dp)
    lib().orc_logpdf_chol_batch(len(mu), pm, pl, px, X.shape[1], X.shape[0], po)
    return out


def logpdf_univariate(mu, vs, x) -> float:
    mu, pm = _d(mu); vs, pv = _d(vs); x, px = _d(x)
    return float(lib().orc_logpdf_univariate(len(mu), pm, pv, px))


def logpdf_sparse(mu, ii, jj, vv, logdet, x) -> float:
    mu, pm = _d(mu); ii, pi = _i(ii); jj, pj = _i(jj); vv, pv = _d(vv); x, px = _d(x)
    return float(lib().orc_logpdf_sparse(len(mu), pm, len(vv), pi, pj, pv, float(logdet), px))


def grad_full_batch(mu, sigma_inv, X) -> np.ndarray:
    mu, pm = _d(mu); P, pp = _d(sigma_inv); X, px = _d(X)
    G = np.empty_like(X); pg = G.ctypes.data_as(_dp)
    lib().orc_grad_full_batch(len(mu), pm, pp, px, X.shape[1], X.shape[0], pg, X.shape[1])
    return G


# ---- tree entry points ----------------------------------------------------------------
def height_to_length(parent, heights) -> np.ndarray:
    parent, pp = _i(parent); h, ph = _d(heights)
    out = np.empty_like(h)
    lib().orc_height_to_length(len(parent), pp, ph, out.ctypes.data_as(_dp))
    return out


def get_branches(parent, values) -> np.ndarray:
    parent, pp = _i(parent); v, pv = _d(values)
    out = np.empty(max(len(parent) - 1, 0))
    _check(lib().orc_get_branches(len(parent), pp, pv, out.ctypes.data_as(_dp)))
    return out


def sum_first_two(v) -> np.ndarray:
    v, pv = _d(v)
    out = np.empty(len(v) - 1)
    lib().orc_sum_first_two(len(v), pv, out.ctypes.data_as(_dp))
    return out


def distances(parent, heights, rates, tH, rMu) -> np.ndarray:
    parent, pp = _i(parent); h, ph = _d(heights); r, pr = _d(rates)
    out = np.empty(len(parent) - 2)
    _check(lib().orc_distances(len(parent), pp, ph, pr, float(tH), float(rMu), out.ctypes.data_as(_dp)))
    return out


def tree_loglik_full_batch(parent, heights, rates, tH, rMu, mu, sigma_inv, logdet, native=False):
    """heights, rates: [batch, n_nodes]; returns (ll[batch], log_jacobian_root_branch[batch])."""
    parent, pp = _i(parent); H, ph = _d(heights); R, pr = _d(rates)
    tH, pt = _d(tH); rMu, pm_ = _d(rMu); mu, pmu = _d(mu); P, pP = _d(sigma_inv)
    B = H.shape[0]
    ll = np.empty(B); lj = np.empty(B)
    _check(lib(native).orc_tree_loglik_full_batch(len(parent), pp, ph, pr, pt, pm_, pmu, pP, float(logdet), B,
                                                  ll.ctypes.data_as(_dp), lj.ctypes.data_as(_dp)))
    return ll, lj


def tree_grad_full(parent, heights, rates, tH, rMu, mu, sigma_inv):
    parent, pp = _i(parent); h, ph = _d(heights); r, pr = _d(rates)
    mu, pmu = _d(mu); P, pP = _d(sigma_inv)
    gh = np.empty_like(h); gr = np.empty_like(r)
    gt = C.c_double(); gm = C.c_double()
    _check(lib().orc_tree_grad_full(len(parent), pp, ph, pr, float(tH), float(rMu), pmu, pP,
                                    gh.ctypes.data_as(_dp), gr.ctypes.data_as(_dp), C.byref(gt), C.byref(gm)))
    return gh, gr, gt.value, gm.value


# ---- prior (prior_oracle.c) -------------------------------------------------------------------
CLOCK_MODELS = {"UncorrelatedGamma": 0, "UncorrelatedLogNormal": 1, "UncorrelatedWhiteNoise": 2, "AutocorrelatedLogNormal": 3}


def compute_de(la, mu, rho, dt, e0, near_critical=False):
    d, e = C.c_double(), C.c_double()
    f = lib().orp_compute_de_near_critical if near_critical else lib().orp_compute_de
    f(la, mu, rho, dt, e0, C.byref(d), C.byref(e))
    return d.value, e.value


def birth_death(cond_mrca, la, mu, rho, parent, lengths) -> float:
    """ln birthDeath; cond_mrca False = ConditionOnTimeOfOrigin (old API: WithStem), True = ...OfMrca (WithoutStem)."""
    parent, pp = _i(parent); ln, pl = _d(lengths)
    return float(lib().orp_birth_death(int(bool(cond_mrca)), la, mu, rho, len(parent), pp, pl))


def relaxed_clock(model, m, v, tlen, rates) -> float:
    t, pt = _d(tlen); r, pr = _d(rates)
    return float(lib().orp_relaxed_clock(CLOCK_MODELS[model] if isinstance(model, str) else int(model), m, v, len(t), pt, pr))


class PriorSpec:
    """Everything `priorFunction ht md cb cs bs` closes over (app/Probability.hs:127-150), on pre-order node ids.
    calibrations: (node, lo or None, lo_p, hi or None, hi_p); constraints: (young, old, p); braces: (nodes, sd)."""

    def __init__(self, parent, ht, model, calibrations=(), constraints=(), braces=()):
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.ht = float(ht)
        self.model = CLOCK_MODELS[model] if isinstance(model, str) else int(model)
        c = list(calibrations)
        self.cal_node = np.array([x[0] for x in c], np.int32)
        self.cal_has_lo = np.array([x[1] is not None for x in c], np.int32)
        self.cal_lo = np.array([x[1] if x[1] is not None else 0.0 for x in c], np.float64)
        self.cal_lo_p = np.array([x[2] if x[1] is not None else 0.0 for x in c], np.float64)
        self.cal_has_hi = np.array([x[3] is not None for x in c], np.int32)
        self.cal_hi = np.array([x[3] if x[3] is not None else 0.0 for x in c], np.float64)
        self.cal_hi_p = np.array([x[4] if x[3] is not None else 0.0 for x in c], np.float64)
        k = list(constraints)
        self.con_young = np.array([x[0] for x in k], np.int32)
        self.con_old = np.array([x[1] for x in k], np.int32)
        self.con_p = np.array([x[2] for x in k], np.float64)
        b = list(braces)
        self.br_ptr = np.concatenate([[0], np.cumsum([len(x[0]) for x in b])]).astype(np.int32)
        self.br_nodes = np.array([n for x in b for n in x[0]], np.int32)
        self.br_sd = np.array([x[1] for x in b], np.float64)


def prior(spec: PriorSpec, birth, death, tH, heights, rMu, rVar, rates):
    """(ln prior, [node priors, birth-death block, relaxed-clock block]) for one state."""
    h, ph = _d(heights); r, pr = _d(rates)
    comp = np.empty(3)
    ip = lambda a: a.ctypes.data_as(_ip)
    dp = lambda a: a.ctypes.data_as(_dp)
    v = lib().orp_prior(spec.ht, spec.model, len(spec.parent), ip(spec.parent), float(birth), float(death), float(tH), ph,
                        float(rMu), float(rVar), pr,
                        len(spec.cal_node), ip(spec.cal_node), ip(spec.cal_has_lo), dp(spec.cal_lo), dp(spec.cal_lo_p),
                        ip(spec.cal_has_hi), dp(spec.cal_hi), dp(spec.cal_hi_p),
                        len(spec.con_young), ip(spec.con_young), ip(spec.con_old), dp(spec.con_p),
                        len(spec.br_sd), ip(spec.br_ptr), ip(spec.br_nodes), dp(spec.br_sd), dp(comp))
    return float(v), comp


# ---- mh_oracle.c: CPU twin of the lock-step Metropolis-Hastings-Green driver ---------------------------------------
def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().orm_philox4x32(c, k, o)
    return [int(x) for x in o]


def uniform_pair(seed, chain, step, d):
    u = np.empty(2)
    lib().orm_uniform_pair(seed, chain, step, d, u.ctypes.data_as(_dp))
    return u


def erfinv(y) -> float:
    return float(lib().orm_erfinv(float(y)))


def tn_logpdf(m, s, a, b, x) -> float:
    return float(lib().orm_tn_logpdf(m, s, a, b, x))


def tn_quantile(m, s, a, b, p) -> float:
    return float(lib().orm_tn_quantile(m, s, a, b, p))


def gamma_draw(seed, chain, step, shape, scale) -> float:
    return float(lib().orm_gamma_draw(seed, chain, step, shape, scale))


class MhModel:
    """Topology + likelihood operands + PriorSpec + proposal table (dict of equally long arrays: kind, node, n1, n2,
    jac_root, dim, p0, p1) for orm_run."""

    def __init__(self, parent, mu, sigma_inv, logdet, spec: PriorSpec, table: dict):
        self.keep = k = {}
        k["parent"] = np.ascontiguousarray(parent, np.int32)
        k["mu"] = np.ascontiguousarray(mu, np.float64)
        k["sigma_inv"] = np.ascontiguousarray(sigma_inv, np.float64)
        for name in ("kind", "node", "n1", "n2", "jac_root", "dim"):
            k[name] = np.ascontiguousarray(table[name], np.int32)
        for name in ("p0", "p1"):
            k[name] = np.ascontiguousarray(table[name], np.float64)
        self.spec = spec
        self.n_nodes = len(k["parent"])
        self.n_prop = len(k["kind"])
        ip = lambda a: a.ctypes.data_as(_ip)
        dp = lambda a: a.ctypes.data_as(_dp)
        self.c = _OrmModel(self.n_nodes, ip(k["parent"]), dp(k["mu"]), dp(k["sigma_inv"]), float(logdet), spec.ht, spec.model,
                           len(spec.cal_node), ip(spec.cal_node), ip(spec.cal_has_lo), dp(spec.cal_lo), dp(spec.cal_lo_p),
                           ip(spec.cal_has_hi), dp(spec.cal_hi), dp(spec.cal_hi_p),
                           len(spec.con_young), ip(spec.con_young), ip(spec.con_old), dp(spec.con_p),
                           len(spec.br_sd), ip(spec.br_ptr), ip(spec.br_nodes), dp(spec.br_sd),
                           self.n_prop, ip(k["kind"]), ip(k["node"]), ip(k["n1"]), ip(k["n2"]), ip(k["jac_root"]), ip(k["dim"]),
                           dp(k["p0"]), dp(k["p1"]))


class MhChains:
    """State of `batch` chains of the CPU twin; same call sequence as the device driver (run / tune / sums)."""

    def __init__(self, model: MhModel, birth, death, tH, H, rMu, rVar, R, seed, chain0=0):
        self.m = model
        f = lambda a: np.array(a, np.float64, order="C")
        self.birth, self.death, self.tH, self.rMu, self.rVar = f(birth), f(death), f(tH), f(rMu), f(rVar)
        self.H, self.R = f(H), f(R)
        self.B = len(self.tH)
        self.seed, self.step, self.chain0 = int(seed), 0, int(chain0)
        P, n = model.n_prop, model.n_nodes
        self.tune = np.ones((self.B, P))
        self.beta = np.ones(self.B)            # reciprocal temperatures (MC3); set_temperatures() of the device driver
        self.acc = np.zeros((self.B, P), np.int32)
        self.tried = np.zeros((self.B, P), np.int32)
        self.post = np.zeros((self.B, 3))
        self.age_sum = np.zeros((self.B, n))
        self.age_sq = np.zeros((self.B, n))
        self.n_samples = 0

    def run(self, schedule, accumulate=False, trace=False):
        sched = np.ascontiguousarray(schedule, np.int32)
        n_iter, S = sched.shape
        ta = np.empty((n_iter * S, self.B)) if trace else None
        tk = np.empty((n_iter * S, self.B), np.int8) if trace else None
        dp = lambda a: a.ctypes.data_as(_dp) if a is not None else None
        ip = lambda a: a.ctypes.data_as(_ip)
        rc = lib().orm_run(C.byref(self.m.c), self.B, dp(self.birth), dp(self.death), dp(self.tH), dp(self.H), dp(self.rMu),
                           dp(self.rVar), dp(self.R), self.H.shape[1], ip(sched), n_iter, S, self.seed, self.step, self.chain0,
                           dp(self.beta), dp(self.tune), ip(self.acc), ip(self.tried), dp(self.post),
                           dp(self.age_sum) if accumulate else None, dp(self.age_sq) if accumulate else None, dp(ta),
                           tk.ctypes.data_as(C.POINTER(C.c_int8)) if trace else None)
        if rc:
            raise OracleError(f"orm_run: {rc}")
        self.step += n_iter * S
        if accumulate:
            self.n_samples += n_iter
        return (ta, tk) if trace else None

    def autotune(self):
        ip = lambda a: a.ctypes.data_as(_ip)
        lib().orm_tune(C.byref(self.m.c), self.B, self.tune.ctypes.data_as(_dp), ip(self.acc), ip(self.tried))


def propose_once(model: MhModel, p, t, seed, chain, step, sc, H, R):
    """Apply proposal row p once: (scalars', H', R', ln q-ratio, ln Jacobian)."""
    sc = np.ascontiguousarray(sc, np.float64); H = np.ascontiguousarray(H, np.float64); R = np.ascontiguousarray(R, np.float64)
    sc1 = np.empty(5); H1 = np.empty_like(H); R1 = np.empty_like(R); q = np.empty(1); j = np.empty(1)
    dp = lambda a: a.ctypes.data_as(_dp)
    lib().orm_propose_once(C.byref(model.c), int(p), float(t), int(seed), int(chain), int(step), dp(sc), dp(H), dp(R), dp(sc1),
                           dp(H1), dp(R1), dp(q), dp(j))
    return sc1, H1, R1, float(q[0]), float(j[0])
