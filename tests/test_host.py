"""CPU tests of the host-side mirror (tree / state / data-file plumbing) and of the C-ABI surface."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import mcmc_date_amd as M
import oracle as O
from mcmc_date_amd import _capi, synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------
# C ABI: the library loads and exports exactly what include/mcmcdate_mvn.h declares
# ---------------------------------------------------------------------------------------------
def header_functions():
    src = open(os.path.join(ROOT, "include", "mcmcdate_mvn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcd_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 15 and "mcd_mvn_logpdf_batch" in names and "mcd_tree_grad_batch" in names
    lib = _capi.lib()                                   # raises if the .so is missing: no fallback
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mcmcdate_mvn.h but not exported"
    assert sorted(_capi.SYMBOLS) == names               # the ctypes table covers the header exactly
    assert b"gfx950" in lib.mcd_version()


def _haskell_imports(path):
    """(symbol, number of arguments) of every `foreign import ccall` of a Haskell source file (function pointers `&sym` skipped)."""
    src = open(path).read()
    src = re.sub(r"--.*", "", src)
    out = []
    for m in re.finditer(r'foreign import ccall\s+(?:safe|unsafe)?\s*"(&?)(mcd_[a-z_0-9]+)"\s*\n?\s*\w+\s*::(.*?)(?=\n\s*\n|\nforeign|\ncheck|\Z)', src, flags=re.S):
        if m.group(1) == "&":
            continue
        sig = m.group(3)
        depth, parts, cur = 0, [], ""
        i = 0
        while i < len(sig):                                  # split on top-level "->"
            c = sig[i]
            if c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            if depth == 0 and sig[i:i + 2] == "->":
                parts.append(cur)
                cur = ""
                i += 2
                continue
            cur += c
            i += 1
        parts.append(cur)
        out.append((m.group(2), len(parts) - 1))
    return out


def test_haskell_bindings_name_real_symbols_with_the_headers_arity():
    """haskell/McmcDate/Gpu.hs (the literal drop-in + raw bindings) and GpuSampler.hs (the replacement of runMetropolisHastingsGreen,
    app/Main.hs:460-479) cannot be compiled here (no GHC); what CAN be checked: every `foreign import ccall` names a symbol the library
    exports, with as many arguments as the header declares -- and the sampler module imports every mcd_mh_* entry point its loop needs."""
    need = {"mcd_mh_create", "mcd_mh_create_sparse", "mcd_mh_destroy", "mcd_mh_set_state", "mcd_mh_get_state", "mcd_mh_run", "mcd_mh_tune",
            "mcd_mh_get_age_sums", "mcd_mh_reset_age_sums", "mcd_mh_mc3_init", "mcd_mh_mc3_swap", "mcd_mh_mc3_get", "mcd_mh_last_path"}
    for name in ("Gpu.hs", "GpuSampler.hs"):
        imps = _haskell_imports(os.path.join(ROOT, "haskell", "McmcDate", name))
        assert len(imps) >= 15, name
        for sym, n_args in imps:
            assert sym in _capi.SYMBOLS, f"{name}: {sym} is not in the C ABI"
            assert n_args == len(_capi.SYMBOLS[sym][1]), f"{name}: {sym} takes {len(_capi.SYMBOLS[sym][1])} arguments, the import has {n_args}"
        if name == "GpuSampler.hs":
            assert need <= {s for s, _ in imps}, need - {s for s, _ in imps}


def test_options_table():
    """mcd_set_option / mcd_get_option: one explicit table instead of getenv on the hot path; unknown names are refused."""
    M.set_option("MCD_MH_SEGMENTS", 0)
    assert M.get_option("MCD_MH_SEGMENTS") == 0
    M.set_option("MCD_MH_SEGMENTS", None)
    assert M.get_option("MCD_MH_SEGMENTS") is None
    with pytest.raises(M.McdError):
        M.set_option("MCD_NO_SUCH_KNOB", 1)
    # every knob the header names is in the table (and the table's names are the header's)
    hdr = open(os.path.join(ROOT, "include", "mcmcdate_mvn.h")).read()
    doc = hdr[hdr.index("Test and tuning knobs"):hdr.index("int mcd_set_option(")]
    names = sorted(set(re.findall(r'"(MCD_[A-Z0-9_]+)"', doc)))
    table = sorted(set(re.findall(r'"(MCD_[A-Z0-9_]+)"', open(os.path.join(ROOT, "mcmc-date_amd", "csrc", "options.cpp")).read())))
    assert names == table, (sorted(set(names) ^ set(table)))
    for nm in names:
        M.set_option(nm, 1)
        assert M.get_option(nm) == 1
        M.set_option(nm, None)
        assert M.get_option(nm) is None
    src = ""
    for f in os.listdir(os.path.join(ROOT, "mcmc-date_amd", "csrc")):
        if f.endswith((".hip", ".hpp", ".cpp", ".h")) and f not in ("options.cpp", "options.h"):
            src += open(os.path.join(ROOT, "mcmc-date_amd", "csrc", f)).read()
    # the only getenv left outside options.cpp: the load-time seed of the process default form (MCD_WIDE)
    assert len(re.findall(r"\bgetenv\(", src)) == 2 and src.count('getenv("MCD_WIDE")') == 2


def test_no_device_is_a_loud_error_not_a_fallback():
    lib = _capi.lib()
    if lib.mcd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(M.NoDevice, match="no CPU path"):
        M.MvnLikelihood.from_covariance(np.zeros(3), np.eye(3))


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "mcmc-date_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt and "mvn_oracle" not in txt, f


# ---------------------------------------------------------------------------------------------
# tree mirror (app/Tools.hs, lib/Mcmc/Tree/Types.hs) against the oracle
# ---------------------------------------------------------------------------------------------
def test_newick_and_branch_order():
    topo, ln = M.parse_newick("((a:0.4,b:0.4)x:0.6,c:1.0):0.0;")
    assert list(topo.parent) == [-1, 0, 1, 1, 0] and topo.names == ["", "x", "a", "b", "c"]
    assert np.array_equal(M.get_branches(topo, ln), [0.6, 1.0, 0.4, 0.4])
    assert np.array_equal(M.sum_first_two(M.get_branches(topo, ln)), [1.6, 0.4, 0.4])
    assert list(M.branch_slots(topo)) == [-1, 0, 1, 2, 0]
    h = np.array([1.0, 0.4, 0.0, 0.0, 0.0])
    assert np.array_equal(M.height_tree_to_length_tree(topo, h), O.height_to_length(topo.parent, h))
    with pytest.raises(M.TreeError, match="not bifurcating"):
        M.get_branches(M.parse_newick("(a:1,b:1,c:1);")[0], np.zeros(4))
    with pytest.raises(M.TreeError):
        M.Topology(np.array([-1, 0, 3, 0], np.int32))          # not pre-order


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_topologies_match_oracle(seed):
    topo = S.random_topology(7 + 5 * seed, seed)
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(topo.n_nodes)
    assert np.array_equal(M.get_branches(topo, v), O.get_branches(topo.parent, v))
    st = S.random_states(topo, 3, seed)
    for b in range(3):
        x = M.State(1.0, 1.0, st.time_height[b], st.heights[b], st.rate_mean[b], 1.0, st.rates[b])
        x.rate_tree = x.rate_tree.copy()
        x.rate_tree[0] = 0.0
        assert x.is_valid(topo)
        d = M.sum_first_two(M.get_branches(topo, M.height_tree_to_length_tree(topo, st.heights[b])) *
                            M.get_branches(topo, st.rates[b])) * (st.time_height[b] * st.rate_mean[b])
        assert np.allclose(d, O.distances(topo.parent, st.heights[b], st.rates[b], st.time_height[b], st.rate_mean[b]),
                           rtol=1e-15)
    bad = M.State(1.0, 1.0, 1.0, np.zeros(topo.n_nodes), 1.0, 1.0, np.ones(topo.n_nodes))
    assert not bad.is_valid(topo)


def test_fixture_trees_are_valid_states(golden):
    fx = golden["12-leaves-variable-rate"]
    topo = M.Topology(fx["parent"], list(fx["names"]))
    assert topo.n_nodes == 23 and int(topo.leaves.sum()) == 12
    for b in (0, 1, 63):
        x = M.State(1.0, 1.0, fx["tH"][b], fx["H"][b], fx["rMu"][b], 1.0, fx["R"][b])
        assert x.is_valid(topo)


# ---------------------------------------------------------------------------------------------
# .data files (app/Main.hs:75-99, 240, 286)
# ---------------------------------------------------------------------------------------------
def test_data_file_round_trip(tmp_path, golden):
    fx = golden["06-leaves-constant-rate"]
    full = M.Full(fx["mu"], fx["sigma_inv"], float(fx["logdet"]))
    p = tmp_path / "t.data"
    M.write_data_file(str(p), full)
    raw = json.load(open(p))
    assert raw["tag"] == "FullS" and len(raw["contents"]) == 3 and len(raw["contents"][1]) == 9
    back = M.read_data_file(str(p))
    assert isinstance(back, M.Full) and np.array_equal(back.sigma_inv, fx["sigma_inv"]) and back.logdet_sigma == float(fx["logdet"])
    for lhd in (M.Univariate(fx["mu"], np.diag(fx["sigma"])), M.Sparse(fx["mu"], [((0, 0), 2.0), ((1, 2), -0.5)], -3.0), M.NoData()):
        M.write_data_file(str(p), lhd)
        assert type(M.read_data_file(str(p))) is type(lhd)
    p.write_text('{"tag": "Bogus"}')
    with pytest.raises(ValueError, match="Could not decode"):
        M.read_data_file(str(p))


def test_synthetic_generators_are_deterministic():
    mu1, s1 = S.random_spd_problem(64, 7)
    mu2, s2 = S.random_spd_problem(64, 7)
    assert np.array_equal(mu1, mu2) and np.array_equal(s1, s2)
    assert np.all(np.linalg.eigvalsh(s1) > 0) and 1e-4 <= np.diag(s1).min() and np.diag(s1).max() <= 1e-3
    X = S.sample_chains(mu1, s1, 8, 7)
    assert X.shape == (8, 64) and np.array_equal(X, S.sample_chains(mu1, s1, 8, 7))
    t = S.random_topology(129, 256)
    assert t.n_nodes == 257 and t.n_nodes - 2 == 255


# ---------------------------------------------------------------------------------------------
# Hamiltonian glue (app/Hamiltonian.hs:33-60)
# ---------------------------------------------------------------------------------------------
def test_hamiltonian_position_vector(golden):
    fx = golden["24-leaves-braces"]
    topo = M.Topology(fx["parent"])
    mask = M.get_mask(True, topo)
    # 2 + [tH] + (L-2) + 2 + (2L-2) = 73 for L = 24 with calibrations
    assert int(mask.sum()) == 73 and int(M.get_mask(False, topo).sum()) == 72
    x = M.State(0.7, 1.3, float(fx["tH"][4]), fx["H"][4], float(fx["rMu"][4]), 0.9, fx["R"][4])
    v = M.to_vector(mask, x)
    assert len(v) == 73
    # reverse fold order: the last rate-tree branch comes first, timeBirthRate last
    assert v[0] == fx["R"][4][-1] and v[-1] == 0.7 and v[-2] == 1.3 and v[-3] == float(fx["tH"][4])
    y = M.from_vector_with(mask, x, v * 2.0)
    assert y.time_birth_rate == 1.4 and y.rate_variance == 1.8 and y.time_tree[0] == x.time_tree[0]   # root height untouched
    assert np.array_equal(y.rate_tree[1:], 2.0 * np.asarray(x.rate_tree)[1:]) and y.rate_tree[0] == x.rate_tree[0]
    assert np.array_equal(np.asarray(y.time_tree)[topo.leaves], np.zeros(24))                        # leaves untouched
    z = M.from_vector_with(mask, y, v)
    assert np.array_equal(M.to_vector(mask, z), v)
    g = M.grad_to_vector(mask, fx["gH"][4], fx["gR"][4], float(fx["gtH"][4]), float(fx["grMu"][4]))
    assert len(g) == 73 and g[0] == fx["gR"][4][-1] and g[-1] == 0.0 and g[-3] == float(fx["gtH"][4])
    with pytest.raises(ValueError):
        M.to_vector(mask[:-1], x)


def test_node_age_summary_follows_the_reference_script():
    """scripts/trees-monitor-summary-ultrametric:149-175: drop round(l * burn-in) samples, ML variance, 95 % interval =
    sorted[floor(0.025 l)] .. sorted[floor(0.025 l) + floor(0.95 l) - 1]."""
    from mcmc_date_amd import monitor as MO

    rng = np.random.default_rng(4)
    ages = rng.gamma(5.0, 2.0, size=(1000, 3))
    s = MO.summarize_node_ages(ages, burn_in=0.25, names=["r", "x", "a"])
    kept = ages[250:]
    l = len(kept)
    for v in range(3):
        srt = sorted(kept[:, v])
        assert abs(s.mean[v] - kept[:, v].mean()) < 1e-12 and abs(s.variance[v] - np.mean((kept[:, v] - kept[:, v].mean()) ** 2)) < 1e-12
        assert s.minimum[v] == srt[0] and s.maximum[v] == srt[-1]
        assert s.ci_lower[v] == srt[int(l * 0.025)] and s.ci_upper[v] == srt[int(l * 0.025) + int(l * 0.95) - 1]
    lines = s.render().splitlines()
    assert lines[0] == "Index\tName\tMean\tVariance\tMin\tMax\t95CILower\t95CIUpper" and lines[2].startswith("1\tx\t") and len(lines) == 4
    # five samples, no burn-in: floor(0.125) = 0, floor(4.75) = 4 -> the interval covers sorted[0..3]
    tiny = MO.summarize_node_ages(np.array([[5.0], [1.0], [3.0], [2.0], [4.0]]), burn_in=0.0)
    assert (tiny.ci_lower[0], tiny.ci_upper[0], tiny.minimum[0], tiny.maximum[0]) == (1.0, 4.0, 1.0, 5.0)
    with pytest.raises(ValueError):
        MO.summarize_node_ages(np.zeros((1, 2)), burn_in=0.9)


class _GaussianLeapfrog:
    """Stand-in for mcmc_date_amd.hmc.Leapfrog with an analytic target (independent normals with given sds): the NUTS
    control flow is host code and can be exercised without a device."""

    def __init__(self, batch, sds, seed=0):
        self.batch, self.dim = batch, len(sds)
        self.sd = np.asarray(sds, float)
        self.q = np.random.default_rng(seed).normal(size=(batch, self.dim)) * self.sd

    def _lp_grad(self, q):
        return -0.5 * np.sum((q / self.sd) ** 2, axis=1), -q / self.sd ** 2

    def position(self):
        lp, g = self._lp_grad(self.q)
        return self.q.copy(), lp, g

    def step_from(self, q, p, grad, eps, inv_mass, direction=None, have_grad=True):
        e = np.asarray(eps, float)[:, None] * (np.ones((self.batch, 1)) if direction is None else np.asarray(direction)[:, None])
        p = p + 0.5 * e * grad
        q = q + e * inv_mass * p
        lp, g = self._lp_grad(q)
        p = p + 0.5 * e * g
        self.q = q.copy()
        return q, p, g, lp


def test_nuts_control_flow_on_an_analytic_target():
    """Hoffman & Gelman's Algorithm 3 + 6 as restated in mcmc_date_amd/hmc.py, on independent normals with very
    different scales: after dual-averaging warm-up the chains reproduce means and variances."""
    from mcmc_date_amd.hmc import DualAveraging, nuts_transition

    sds = np.array([0.2, 1.0, 3.0, 0.5])
    B = 32
    lf = _GaussianLeapfrog(B, sds, seed=1)
    rng = np.random.default_rng(2)
    inv_mass = np.ones(4)                                   # unit masses: the step size must adapt to the smallest scale
    da = DualAveraging(np.full(B, 0.05), delta=0.65)
    eps = np.full(B, 0.05)
    for _ in range(200):
        alpha, depth = nuts_transition(lf, rng, eps, inv_mass, max_depth=7)
        eps = da.update(alpha)
    eps = da.final()
    assert np.all((eps > 0.04) & (eps < 0.8)), eps          # stability limit 2 * min(sd) = 0.4; delta = 0.65 sits around it
    draws, alphas, depths = [], [], []
    for _ in range(250):
        alpha, depth = nuts_transition(lf, rng, eps, inv_mass, max_depth=7)
        alphas.append(alpha.mean())
        depths.append(depth.mean())
        draws.append(lf.q.copy())
    x = np.array(draws).reshape(-1, 4)
    assert 0.5 < np.mean(alphas) < 0.95, np.mean(alphas)   # the averaged step size is a little conservative after a short warm-up
    assert np.all(np.abs(x.mean(axis=0)) < 0.06 * sds * 4), x.mean(axis=0) / sds
    assert np.all(np.abs(x.std(axis=0) / sds - 1.0) < 0.08), x.std(axis=0) / sds
    assert 2.5 < np.mean(depths) <= 7.0                     # the widest coordinate needs long trajectories


def test_split_schedule_selftest():
    """The schedule of the row-split form (csrc/host_factor.cpp: build_split_schedule) is host code: the library walks it on
    the CPU for a random residual vector -- tile runs, whole and cut row blocks, the fixed-order combination, the tile pairs
    as the kernel loads them -- and must find |W r|^2.  No GPU involved."""
    import ctypes as C

    L = C.CDLL(M._capi.LIB_PATH)
    f = L.mcd_split_schedule_selftest_
    f.restype = C.c_double
    f.argtypes = [C.c_int, C.c_int, C.c_uint]
    for n in (1, 15, 16, 17, 129, 160, 192, 193, 255, 256, 257, 300, 384, 500, 512, 513, 700, 768, 1000, 1021, 1023, 1024):
        for G in (8, 16, 32):
            err = f(n, G, 7 * n + G)
            assert 0.0 <= err <= 1e-14, (n, G, err)


def test_load_calibrations_from_tree(tmp_path):
    """`loadCalibrationsFromTree` (lib/Mcmc/Tree/Prior/Node/CalibrationFromTree.hs:119-130) on the reference's own
    mtCDNApri analysis (bench/comparison_with_mcmctree/02_McmcDate/01_McmcDate/data/mtCDNApri_MD.trees; inputs kept in
    tests/golden/mtCDNApri_prior_samples.json): MCMCtree's B / U labels on a Newick tree with quoted labels, nodes found as the
    MRCA of the labelled node's leftmost and rightmost leaf -- the pre-order indices 0, 1, 3 the reference's own result table
    names (README.md:641-645) --, default probability mass 0.01, the L form with its ignored Cauchy parameters, no label at all."""
    import json

    from mcmc_date_amd.prepare import prepare

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "mtCDNApri_prior_samples.json")))
    paths = {}
    for k in ("rooted_tree", "calibration_tree", "tree_list"):
        paths[k] = str(tmp_path / k)
        open(paths[k], "w").write(fx["inputs"][k])
    prep = prepare(paths["tree_list"], paths["rooted_tree"], "NoLikelihood")
    topo = prep.topology
    assert [i for i in range(topo.n_nodes) if not topo.leaves[i]] == fx["nodes"] == [0, 1, 2, 3, 5, 9]
    cal = M.load_calibrations_from_tree(topo, paths["calibration_tree"])
    got = sorted((c.node, c.name, c.lower, c.lower_p, c.upper, c.upper_p) for c in cal)
    assert got == [(0, "human-gibbon", None, 0.0, 100.0, 0.025), (1, "human-sumatran", 12.0, 0.025, 16.0, 0.025),
                   (3, "human-bonobo", 6.0, 0.025, 8.0, 0.025)]
    assert M.get_mean_root_height(cal) == 50.0                      # getMeanRootHeight: (0 + 100) / 2
    other = str(tmp_path / "other.tree")
    open(other, "w").write("(((human,(chimpanzee,bonobo))'L(5)',gorilla)'L(9,0.1,1.0,0.05)',(orangutan,sumatran)'U(20)',gibbon)'B(10,90)';")
    got = sorted((c.node, c.lower, c.lower_p, c.upper, c.upper_p) for c in M.load_calibrations_from_tree(topo, other))
    assert got == [(0, 10.0, 0.01, 90.0, 0.01), (2, 9.0, 0.05, None, 0.0), (3, 5.0, 0.01, None, 0.0), (9, None, 0.0, 20.0, 0.01)]
    with pytest.raises(ValueError):
        M.load_calibrations_from_tree(topo, paths["rooted_tree"])      # no calibrations found


def test_python_philox_mirror_matches_the_oracle():
    """sampler.philox4x32 / uniform_pair (the host-side mirror of csrc/mh_device.hpp: philox_block, used by mc3_swap_host) against
    Random123's known answers and the oracle's generator."""
    import oracle as O
    from mcmc_date_amd import sampler as SM

    assert list(SM.philox4x32([0, 0, 0, 0], [0, 0])) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(SM.philox4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    for seed, chain, step, d in [(3, 7, 11, 5), (2 ** 40 + 17, 4000, 2 ** 33 + 5, 0xFFFFFFFF), (SM.MC3_STREAM_DOMAIN ^ 5, 12, 0, 2)]:
        assert tuple(O.uniform_pair(seed, chain, step, d)) == SM.uniform_pair(seed, chain, step, d)


def test_mc3_swap_host_keeps_a_permutation_per_group():
    from mcmc_date_amd import sampler as SM

    rng = np.random.default_rng(1)
    ladder = np.array([1.0, 0.8, 0.5, 0.3, 0.1])
    rank = (np.arange(40) % 5).astype(np.int32)
    tried, acc = np.zeros(4, np.int64), np.zeros(4, np.int64)
    for phase in range(50):
        SM.mc3_swap_host(rank, rng.normal(size=40) * 3, ladder, 4, 9, phase, tried, acc)
        assert all(sorted(rank[g * 5:(g + 1) * 5].tolist()) == [0, 1, 2, 3, 4] for g in range(8))
    assert tried.sum() == 50 * 8 * 4 and np.all(tried == 400) and 0 < acc.sum() < tried.sum()      # n_swaps = n - 1: every pair once per phase
    r2 = rank.copy()
    SM.mc3_swap_host(r2, np.full(40, np.nan), ladder, 4, 9, 99)                                  # NaN posteriors: no swap
    assert np.array_equal(r2, rank)
