// mcmcdate.hpp -- C++ host-side mirror of the reference's likelihood plugin surface, on top of the
// C ABI (include/mcmcdate_mvn.h).  Header-only.  Names and argument meaning follow the reference:
//
//   LikelihoodData = Full | Sparse | Univariate | NoData            app/Probability.hs:210-235
//   likelihoodFunction :: LikelihoodData -> LikelihoodFunction I    app/Probability.hs:277-281
//   jacobianRootBranch                                              app/Probability.hs:408-410
//   getBranches / sumFirstTwo                                       app/Tools.hs:36-48
//   heightTreeToLengthTree                                          lib/Mcmc/Tree/Types.hs:224-233
//   I (state record)                                                app/State.hs:70-91
//   priorFunction ht md cb cs bs                                    app/Probability.hs:127-150
//   initWith, weightNBranches, proposals (the Metropolis-Hastings cycle)   app/Definitions.hs:96-130, 145-278
//   the lock-step many-chain driver over that cycle (mhg of package `mcmc`, app/Main.hs:460-479)
//
// The reference is Haskell; its toolchain is absent from this image, so the host layer above the
// C ABI is written in C++ (compiled code, like the reference) -- haskell/McmcDate/Gpu.hs holds the
// source-level FFI shim a maintainer would add to the reference itself.  Structural faults throw
// std::runtime_error (the reference calls `error`); numeric NaN/Inf flow through the returned value.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <random>
#include <stdexcept>
#include <optional>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/mcmcdate_mvn.h"

namespace mcmcdate {

using Vec = std::vector<double>;

// ---- trees: pre-order parent arrays -------------------------------------------------------------
struct Topology {
    std::vector<int32_t> parent;   // parent[0] = -1, every node before its children
    int nNodes() const { return (int)parent.size(); }
    std::pair<int, int> rootChildren() const
    {
        int l = -1, r = -1, n = 0;
        for (int v = 1; v < nNodes(); ++v)
            if (parent[v] == 0) { (n == 0 ? l : r) = v; ++n; }
        if (n != 2) throw std::runtime_error("getBranches: Root node is not bifurcating.");   // app/Tools.hs:43
        return {l, r};
    }
};

// getBranches -- app/Tools.hs:36-43 (values indexed by pre-order node; the stem is ignored)
inline Vec getBranches(const Topology& t, const Vec& values)
{
    auto [l, r] = t.rootChildren();
    Vec out{values[l], values[r]};
    for (int v = l + 1; v < r; ++v) out.push_back(values[v]);
    for (int v = r + 1; v < t.nNodes(); ++v) out.push_back(values[v]);
    return out;
}
// sumFirstTwo -- app/Tools.hs:47-48
inline Vec sumFirstTwo(const Vec& v)
{
    Vec out{v.at(0) + v.at(1)};
    out.insert(out.end(), v.begin() + 2, v.end());
    return out;
}
// heightTreeToLengthTree -- lib/Mcmc/Tree/Types.hs:224-233
inline Vec heightTreeToLengthTree(const Topology& t, const Vec& heights)
{
    Vec out(heights.size());
    for (int v = 0; v < t.nNodes(); ++v) out[v] = (t.parent[v] < 0 ? heights[v] : heights[t.parent[v]]) - heights[v];
    return out;
}

// ---- the state record `I` -- app/State.hs:70-91 ---------------------------------------------------
struct I {
    double timeBirthRate = 1.0, timeDeathRate = 1.0, timeHeight = 1.0;
    Vec timeTree;                  // HeightTree: relative node heights (pre-order), leaves 0, root 1
    double rateMean = 1.0, rateVariance = 1.0;
    Vec rateTree;                  // LengthTree: relative branch rates (pre-order), [0] = stem (unused)
};

// ---- LikelihoodData -- app/Probability.hs:210-235 -------------------------------------------------
struct Full { Vec mu; Vec sigmaInv; double logDetSigma; };                       // sigmaInv row-major n x n
struct Sparse { Vec mu; std::vector<std::pair<std::pair<int, int>, double>> sigmaInvAssoc; double logDetSigma; };
struct Univariate { Vec mu; Vec vs; };
struct NoData {};
using LikelihoodData = std::variant<Full, Sparse, Univariate, NoData>;

// mcd_mvn_create refuses a matrix that has no Cholesky factor (MCD_ERR_NOT_SPD); likelihoodFunction then takes the product form
struct NotPositiveDefinite : std::runtime_error { using std::runtime_error::runtime_error; };

namespace detail {
inline void check(int rc)
{
    if (rc == MCD_ERR_NOT_SPD) throw NotPositiveDefinite(std::string(mcd_last_error()));
    if (rc != MCD_OK) throw std::runtime_error(std::string(mcd_last_error()));
}
struct MvnDeleter { void operator()(mcd_mvn_t* p) const { mcd_mvn_destroy(p); } };
struct TreeDeleter { void operator()(mcd_tree_t* p) const { mcd_tree_destroy(p); } };
}  // namespace detail

// The library's test and tuning knobs (mcd_set_option: one explicit table instead of environment variables on the hot path; no counterpart
// in the reference).  An empty optional puts a knob back to its default; getOption returns what is set.
inline void setOption(const std::string& name, std::optional<int> value)
{
    const std::string v = value ? std::to_string(*value) : std::string();
    detail::check(mcd_set_option(name.c_str(), value ? v.c_str() : nullptr));
}
inline std::optional<int> getOption(const std::string& name)
{
    int v = 0, is_set = 0;
    detail::check(mcd_get_option(name.c_str(), &is_set, &v));
    return is_set ? std::optional<int>(v) : std::nullopt;
}

// Kernel form of the log-density entry points: column sweep (latency form), matrix-core multiply (throughput form), or
// by dimension and batch size (default).  No counterpart in the reference (one CPU code path); returns the previous form.
enum class LogpdfForm : int { Auto = MCD_FORM_AUTO, Sweep = MCD_FORM_SWEEP, Multiply = MCD_FORM_MULTIPLY };
inline LogpdfForm setLogpdfForm(LogpdfForm f)
{
    const int prev = mcd_set_logpdf_form((int)f);
    if (prev < 0) detail::check(prev);
    return (LogpdfForm)prev;
}


// The closure built by getLikelihoodFunction (app/Main.hs:333-347): operands staged once on a GPU.
class Likelihood {
public:
    Likelihood(const LikelihoodData& lhd, const Topology& topo, int device = 0) : topo_(topo)
    {
        mcd_mvn_t* h = nullptr;
        if (auto* f = std::get_if<Full>(&lhd)) {
            detail::check(mcd_mvn_create(&h, (int)f->mu.size(), f->mu.data(), f->sigmaInv.data(), MCD_MAT_SIGMA_INV, f->logDetSigma, device));
        } else if (auto* s = std::get_if<Sparse>(&lhd)) {
            const size_t n = s->mu.size();
            Vec dense(n * n, 0.0);                         // L.mkSparse, app/Main.hs:95
            for (auto& e : s->sigmaInvAssoc) dense[(size_t)e.first.first * n + e.first.second] += e.second;
            detail::check(mcd_mvn_create(&h, (int)n, s->mu.data(), dense.data(), MCD_MAT_SIGMA_INV, s->logDetSigma, device));
        } else if (auto* u = std::get_if<Univariate>(&lhd)) {
            const size_t n = u->mu.size();
            Vec diag(n * n, 0.0);                          // Sigma = diag vs; logdet = sum (log vs), :274
            for (size_t i = 0; i < n; ++i) diag[i * n + i] = u->vs[i];
            detail::check(mcd_mvn_create(&h, (int)n, u->mu.data(), diag.data(), MCD_MAT_SIGMA, 0.0, device));
        } else {
            topo_.rootChildren();
            return;                                        // NoData: likelihood 1.0 (:281)
        }
        mvn_.reset(h);
        mcd_tree_t* t = nullptr;
        detail::check(mcd_tree_create(&t, h, topo.nNodes(), topo.parent.data()));
        tree_.reset(t);
    }

    // LikelihoodFunction I: one state in, log-likelihood (log domain) out
    double operator()(const I& x) const
    {
        if (!mvn_) return 0.0;
        double ll = 0.0;
        detail::check(mcd_tree_loglik_batch(tree_.get(), x.timeTree.data(), x.rateTree.data(), topo_.nNodes(), &x.timeHeight,
                                            &x.rateMean, 1, 0, nullptr, &ll, nullptr));
        return ll;
    }
    // jacobianRootBranch x = log (1 / rootBranch x)
    double jacobianRootBranch(const I& x) const
    {
        if (!mvn_) throw std::runtime_error("jacobianRootBranch: no likelihood data bound");
        double ll = 0.0, lj = 0.0;
        detail::check(mcd_tree_loglik_batch(tree_.get(), x.timeTree.data(), x.rateTree.data(), topo_.nNodes(), &x.timeHeight,
                                            &x.rateMean, 1, 0, nullptr, &ll, &lj));
        return lj;
    }
    // many chains at once (chain-major host arrays)
    void batch(const double* heights, const double* rates, const double* tH, const double* rMu, int64_t n, double* ll,
               double* logJac = nullptr) const
    {
        detail::check(mcd_tree_loglik_batch(tree_.get(), heights, rates, topo_.nNodes(), tH, rMu, n, 0, nullptr, ll, logJac));
    }
    // logDensityFullMultivariateNormal mu (sigmaInv, logdet) xs  -- app/Probability.hs:166-173
    double logDensity(const Vec& xs) const
    {
        double ll = 0.0;
        detail::check(mcd_mvn_logpdf(mvn_.get(), xs.data(), &ll));
        return ll;
    }

    const mcd_tree_t* treeHandle() const { return tree_.get(); }
    const Topology& topology() const { return topo_; }

private:
    Topology topo_;
    std::unique_ptr<mcd_mvn_t, detail::MvnDeleter> mvn_;
    std::unique_ptr<mcd_tree_t, detail::TreeDeleter> tree_;
};

// likelihoodFunction (Sparse mu sigmaInvSparse logDetSigma) with the precision matrix kept SPARSE on the device (mcd_sparse_*; N up to
// MCD_MAX_SPARSE_DIM): logDensitySparseMultivariateNormal, app/Probability.hs:178-184 -- the reference's route for thousands of branches.
class SparseLikelihood {
public:
    SparseLikelihood(const Sparse& s, const Topology& topo, int device = 0) : topo_(topo)
    {
        std::vector<int32_t> row, col;
        Vec val;
        for (auto& e : s.sigmaInvAssoc) { row.push_back(e.first.first); col.push_back(e.first.second); val.push_back(e.second); }
        mcd_sparse_t* h = nullptr;
        detail::check(mcd_sparse_create(&h, (int)s.mu.size(), s.mu.data(), (int64_t)val.size(), row.data(), col.data(), val.data(), s.logDetSigma, device));
        sp_.reset(h, [](mcd_sparse_t* p) { mcd_sparse_destroy(p); });
        mcd_sparse_tree_t* t = nullptr;
        detail::check(mcd_sparse_tree_create(&t, h, topo.nNodes(), topo.parent.data()));
        tree_.reset(t, [](mcd_sparse_tree_t* p) { mcd_sparse_tree_destroy(p); });
    }
    double operator()(const I& x) const
    {
        double ll = 0.0;
        detail::check(mcd_sparse_tree_loglik_batch(tree_.get(), x.timeTree.data(), x.rateTree.data(), topo_.nNodes(), &x.timeHeight, &x.rateMean, 1, 0,
                                                   nullptr, &ll, nullptr));
        return ll;
    }
    double logDensity(const Vec& xs) const
    {
        double ll = 0.0;
        detail::check(mcd_sparse_logpdf_batch(sp_.get(), xs.data(), (int64_t)xs.size(), 1, 0, nullptr, &ll));
        return ll;
    }

    const Topology& topology() const { return topo_; }
    const mcd_sparse_tree_t* treeHandle() const { return tree_.get(); }

private:
    Topology topo_;
    std::shared_ptr<mcd_sparse_t> sp_;
    std::shared_ptr<mcd_sparse_tree_t> tree_;
};

// likelihoodFunction :: LikelihoodData -> LikelihoodFunction I   (app/Probability.hs:277-281)
inline std::function<double(const I&)> likelihoodFunction(const LikelihoodData& lhd, const Topology& topo, int device = 0)
{
    try {
        auto lik = std::make_shared<Likelihood>(lhd, topo, device);
        return [lik](const I& x) { return (*lik)(x); };
    } catch (const NotPositiveDefinite&) {
        // The dense kernels need a factor; the reference evaluates dx . (P dx) with whatever P the record holds
        // (app/Probability.hs:169, 183): an indefinite precision matrix takes the product form on the device (mcd_sparse_*).
        Sparse sp;
        if (const Full* f = std::get_if<Full>(&lhd)) {
            const int n = (int)f->mu.size();
            sp.mu = f->mu;
            sp.logDetSigma = f->logDetSigma;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (f->sigmaInv[(size_t)i * n + j] != 0.0) sp.sigmaInvAssoc.push_back({{i, j}, f->sigmaInv[(size_t)i * n + j]});
        } else if (const Sparse* s = std::get_if<Sparse>(&lhd)) {
            sp = *s;
        } else {
            throw;
        }
        auto lik = std::make_shared<SparseLikelihood>(sp, topo, device);
        return [lik](const I& x) { return (*lik)(x); };
    }
}

// ---- priorFunction -- app/Probability.hs:127-150 ----------------------------------------------------------------------
enum class RelaxedMolecularClockModel { UncorrelatedGamma = 0, UncorrelatedLogNormal = 1, UncorrelatedWhiteNoise = 2, AutocorrelatedLogNormal = 3 };
// node = pre-order index; lower / upper: absolute ages, hasLower / hasUpper false = Zero / Infinity (Calibration.hs:108-121)
struct Calibration { std::string name; int node; bool hasLower; double lower, lowerP; bool hasUpper; double upper, upperP; };
struct Constraint { std::string name; int young, old; double p; };
struct Brace { std::string name; std::vector<int> nodes; double sd; };

namespace detail {
struct PriorDeleter { void operator()(mcd_prior_t* p) const { mcd_prior_destroy(p); } };
struct MhDeleter { void operator()(mcd_mh_t* p) const { mcd_mh_destroy(p); } };
}  // namespace detail

class PriorFunction {
public:
    PriorFunction(double ht, RelaxedMolecularClockModel md, const std::vector<Calibration>& cb, const std::vector<Constraint>& cs,
                  const std::vector<Brace>& bs, const Topology& topo, int device = 0)
        : topo_(topo)
    {
        std::vector<int32_t> cn, chl, chh, cy, co, bp{0}, bn;
        Vec cl, clp, ch, chp, cp, bsd;
        for (auto& c : cb) {
            cn.push_back(c.node); chl.push_back(c.hasLower); cl.push_back(c.hasLower ? c.lower : 0.0); clp.push_back(c.hasLower ? c.lowerP : 0.0);
            chh.push_back(c.hasUpper); ch.push_back(c.hasUpper ? c.upper : 0.0); chp.push_back(c.hasUpper ? c.upperP : 0.0);
        }
        for (auto& k : cs) { cy.push_back(k.young); co.push_back(k.old); cp.push_back(k.p); }
        for (auto& b : bs) {
            for (int v : b.nodes) bn.push_back(v);
            bp.push_back((int32_t)bn.size());
            bsd.push_back(b.sd);
        }
        mcd_prior_t* p = nullptr;
        detail::check(mcd_prior_create(&p, topo.nNodes(), topo.parent.data(), ht, (int)md, (int)cb.size(), cn.data(), chl.data(), cl.data(),
                                       clp.data(), chh.data(), ch.data(), chp.data(), (int)cs.size(), cy.data(), co.data(), cp.data(),
                                       (int)bs.size(), bp.data(), bn.data(), bsd.data(), device));
        prior_.reset(p);
    }
    // PriorFunction I: log prior of one state
    double operator()(const I& x) const
    {
        double lp = 0.0;
        detail::check(mcd_prior_logprior_batch(prior_.get(), &x.timeBirthRate, &x.timeDeathRate, &x.timeHeight, x.timeTree.data(), &x.rateMean,
                                               &x.rateVariance, x.rateTree.data(), topo_.nNodes(), 1, 0, nullptr, &lp, nullptr));
        return lp;
    }
    const mcd_prior_t* handle() const { return prior_.get(); }

private:
    Topology topo_;
    std::unique_ptr<mcd_prior_t, detail::PriorDeleter> prior_;
};

// ---- initWith -- app/Definitions.hs:96-123 -----------------------------------------------------------------------------
inline I initWith(const Topology& t, const Vec& lengths)
{
    const int n = t.nNodes();
    Vec ln(lengths);
    double sum = 0.0;
    for (int v = 1; v < n; ++v) sum += ln[v];
    const double avg = sum / (n - 1);
    for (int v = 1; v < n; ++v)
        if (ln[v] == 0.0) ln[v] = avg;
    ln[0] = 0.0;
    std::vector<char> leaf(n, 1);
    for (int v = 1; v < n; ++v) leaf[t.parent[v]] = 0;
    Vec dist(n, 0.0);
    double height = 0.0;
    for (int v = 1; v < n; ++v) dist[v] = dist[t.parent[v]] + ln[v];
    for (int v = 0; v < n; ++v)
        if (leaf[v]) height = std::max(height, dist[v]);
    I x;
    x.timeTree.assign(n, 0.0);
    x.rateTree.assign(n, 1.0);
    for (int v = 0; v < n; ++v) x.timeTree[v] = leaf[v] ? 0.0 : (height - dist[v]) / height;
    x.timeTree[0] = 1.0;
    x.rateTree[0] = 0.0;
    return x;
}

// ---- proposals -- app/Definitions.hs:127-278 ----------------------------------------------------------------------------
struct Proposal {
    std::string name;
    int kind = 0, node = 0;
    double p0 = 1.0, p1 = 0.0;
    int n1 = 0, n2 = 0;
    bool jacRoot = false;   // liftProposalWith jacobianRootBranch
    int dim = 1, weight = 1;
};

inline int weightNBranches(int n) { return (int)std::floor(std::log((double)n) / std::log(1.3)); }   // :127-130

// The cycle of `proposals bs calibrationsAvailable x Nothing` in the reference's order (kinds: MCD_PROP_*).
inline std::vector<Proposal> proposals(const Topology& topo, const std::vector<Brace>& braces, bool calibrationsAvailable,
                                       bool exactJacobians = false)
{
    const int n = topo.nNodes();
    std::vector<int> size(n, 1), inner(n, 0), levels(n, 1), plen(n, 0), nch(n, 0);
    std::vector<char> leaf(n, 1);
    for (int v = 1; v < n; ++v) { leaf[topo.parent[v]] = 0; plen[v] = plen[topo.parent[v]] + 1; nch[topo.parent[v]]++; }
    for (int v = 0; v < n; ++v) inner[v] = leaf[v] ? 0 : 1;
    for (int v = n - 1; v > 0; --v) {
        const int p = topo.parent[v];
        size[p] += size[v];
        inner[p] += inner[v];
        levels[p] = std::max(levels[p], levels[v] + 1);
    }
    const int w = weightNBranches(n);
    auto [l, r] = topo.rootChildren();
    std::vector<Proposal> ps;
    auto add = [&](std::string name, int kind, int node, double p0, double p1, int n1, int n2, bool jac, int dim, int weight) {
        ps.push_back(Proposal{std::move(name), kind, node, p0, p1, n1, n2, jac, dim, weight});
    };
    auto subW = [&](int v) { return std::min(3 + levels[v] - 2, 8); };
    add("Time birth rate", MCD_PROP_SCALE_SCALAR, 0, 10.0, 0, 0, 0, false, 1, w);
    add("Time death rate", MCD_PROP_SCALE_SCALAR, 1, 10.0, 0, 0, 0, false, 1, w);
    add("Rate mean", MCD_PROP_SCALE_SCALAR, 3, 10.0, 0, 0, 0, false, 1, w);
    add("Rate variance", MCD_PROP_SCALE_SCALAR, 4, 10.0, 0, 0, 0, false, 1, w);
    const int nInner = inner[0];
    if (nInner - 1 < 1) throw std::runtime_error("scaleRatesAndTreeContrarilyPFunction: no internal nodes to scale");
    add("Rates and time tree", MCD_PROP_SCALE_RATES_TREE_CONTRA, 0, 0.1, 0, nInner - 1, 0, true, nInner - 1 + 2, w);
    auto group = [&](bool atRoot) { return [&, atRoot](int v) { return atRoot ? plen[v] == 1 : plen[v] > 1; }; };
    auto timePs = [&](bool atRoot, const std::string& tag) {
        auto hn = group(atRoot);
        for (int v = 0; v < n; ++v)
            if (!leaf[v] && hn(v)) add(tag + " Time tree node " + std::to_string(v), MCD_PROP_SLIDE_NODE, v, 0.01, 0, 0, 0, atRoot, 1, 5);
        for (int v = 0; v < n; ++v)
            if (!leaf[v] && hn(v)) add(tag + " Time tree node " + std::to_string(v), MCD_PROP_SCALE_SUBTREE_TIME, v, 0.01, 0, inner[v], 0, atRoot, inner[v], subW(v));
    };
    if (!leaf[l] && !leaf[r]) add("[R] Time tree", MCD_PROP_PULLEY, 0, 0.01, 0, inner[l], inner[r], true, inner[l] + inner[r], 6);
    timePs(true, "[R]");
    timePs(false, "[O]");
    for (size_t i = 0; i < braces.size(); ++i)
        add("[B] Time tree " + braces[i].name, MCD_PROP_SLIDE_BRACE, (int)i, 0.01, 0, 0, 0, false, (int)braces[i].nodes.size(), 5);
    add("[R] Rate mean, Rate tree", MCD_PROP_SCALE_NORM_TREE, 3, 100.0, 0, 0, 0, true, n, w);
    add("[R] Rate variance, Rate tree", MCD_PROP_SCALE_VAR_TREE, 0, 100.0, exactJacobians ? 1.0 : 0.0, 0, 0, true, n, w);
    add("[R] Rate variance, Rate tree (autocorrelated)", MCD_PROP_SCALE_VAR_TREE_AUTO, 0, 100.0, 0, 0, 0, true, n, w);
    auto ratePs = [&](bool atRoot, const std::string& tag) {
        auto hn = group(atRoot);
        for (int v = 0; v < n; ++v)
            if (hn(v)) add(tag + " Rate tree branch " + std::to_string(v), MCD_PROP_SCALE_BRANCH_RATE, v, 100.0, 0, 0, 0, atRoot, 1, 3);
        for (int v = 0; v < n; ++v)
            if (!leaf[v] && hn(v)) add(tag + " Rate tree node " + std::to_string(v), MCD_PROP_SCALE_SUBTREE_RATE, v, 100.0, 0, size[v], 0, atRoot, size[v], subW(v));
    };
    ratePs(true, "[R]");
    ratePs(false, "[O]");
    auto contraPs = [&](bool atRoot, const std::string& tag) {
        auto hn = group(atRoot);
        for (int v = 0; v < n; ++v)
            if (!leaf[v] && hn(v)) add(tag + " Trees node " + std::to_string(v), MCD_PROP_SLIDE_NODE_CONTRA, v, 0.1, 0, 0, 0, atRoot, 1 + 1 + nch[v], subW(v));
        for (int v = 0; v < n; ++v)
            if (!leaf[v] && hn(v)) add(tag + " Trees node " + std::to_string(v), MCD_PROP_SCALE_SUBTREE_CONTRA, v, 0.1, 0, inner[v], size[v], atRoot, inner[v] + size[v], subW(v));
    };
    contraPs(true, "[C] [R]");
    contraPs(false, "[C] [O]");
    for (size_t i = 0; i < braces.size(); ++i) {
        int daughters = 0;
        for (int x : braces[i].nodes) daughters += nch[x];
        add("[C] [B] Trees " + braces[i].name, MCD_PROP_SLIDE_BRACE_CONTRA, (int)i, 0.1, 0, 0, 0, false, 2 * (int)braces[i].nodes.size() + daughters, 5);
    }
    if (calibrationsAvailable) {
        add("Time height", MCD_PROP_SCALE_SCALAR, 2, 3000.0, 0, 0, 0, false, 1, w);
        add("Time height, rate mean", MCD_PROP_SCALE_CONTRARILY, 0, 10.0, 0.1, 0, 0, false, 2, w);
        add("[R] Time height, Rate tree", MCD_PROP_SCALE_NORM_TREE, 2, 100.0, 0, 0, 0, true, n, w);
        add("[R] Trees", MCD_PROP_SLIDE_ROOT_CONTRA, 0, 10.0, 0, exactJacobians ? nInner - 1 : nInner, 0, true, 1 + nInner + 2, w);
    }
    return ps;
}

// One iteration of `mcmc`'s default cycle order: every proposal `weight` times, freshly shuffled.
template <class Rng>
inline std::vector<int32_t> cycleSchedule(const std::vector<Proposal>& ps, int nIter, Rng& rng)
{
    std::vector<int32_t> base;
    for (size_t i = 0; i < ps.size(); ++i) base.insert(base.end(), ps[i].weight, (int32_t)i);
    std::vector<int32_t> out;
    for (int it = 0; it < nIter; ++it) {
        std::shuffle(base.begin(), base.end(), rng);
        out.insert(out.end(), base.begin(), base.end());
    }
    return out;
}

// B chains in lock step on one GPU (mhg of package `mcmc`, one state per call, becomes many chains per call).
class Sampler {
public:
    Sampler(const Likelihood& lik, const PriorFunction& prior, std::vector<Proposal> table, int64_t batch, uint64_t seed)
        : topo_(lik.topology()), table_(std::move(table)), batch_(batch)
    {
        std::vector<int32_t> kind, node, n1, n2, jac, dim;
        Vec p0, p1;
        for (auto& p : table_) {
            kind.push_back(p.kind); node.push_back(p.node); n1.push_back(p.n1); n2.push_back(p.n2); jac.push_back(p.jacRoot); dim.push_back(p.dim);
            p0.push_back(p.p0); p1.push_back(p.p1);
        }
        mcd_mh_t* m = nullptr;
        detail::check(mcd_mh_create(&m, lik.treeHandle(), prior.handle(), (int)table_.size(), kind.data(), node.data(), n1.data(), n2.data(),
                                    jac.data(), dim.data(), p0.data(), p1.data(), batch, seed));
        mh_.reset(m);
        stepsPerIteration_ = 0;
        for (auto& p : table_) stepsPerIteration_ += p.weight;
    }
    // the same sampler over a likelihood whose precision matrix stays sparse on the device (likelihoodFunction (Sparse ...),
    // app/Probability.hs:279): trees of 3 .. 2048 nodes (mcd_mh_create_sparse)
    Sampler(const SparseLikelihood& lik, const PriorFunction& prior, std::vector<Proposal> table, int64_t batch, uint64_t seed)
        : topo_(lik.topology()), table_(std::move(table)), batch_(batch)
    {
        std::vector<int32_t> kind, node, n1, n2, jac, dim;
        Vec p0, p1;
        for (auto& p : table_) {
            kind.push_back(p.kind); node.push_back(p.node); n1.push_back(p.n1); n2.push_back(p.n2); jac.push_back(p.jacRoot); dim.push_back(p.dim);
            p0.push_back(p.p0); p1.push_back(p.p1);
        }
        mcd_mh_t* m = nullptr;
        detail::check(mcd_mh_create_sparse(&m, lik.treeHandle(), prior.handle(), (int)table_.size(), kind.data(), node.data(), n1.data(), n2.data(),
                                           jac.data(), dim.data(), p0.data(), p1.data(), batch, seed));
        mh_.reset(m);
        stepsPerIteration_ = 0;
        for (auto& p : table_) stepsPerIteration_ += p.weight;
    }
    void setInitialState(const I& x)   // every chain starts from the same state, like the reference's single chain
    {
        const int nn = topo_.nNodes();
        Vec b(batch_, x.timeBirthRate), d(batch_, x.timeDeathRate), t(batch_, x.timeHeight), m(batch_, x.rateMean), v(batch_, x.rateVariance), H, R;
        for (int64_t c = 0; c < batch_; ++c) { H.insert(H.end(), x.timeTree.begin(), x.timeTree.end()); R.insert(R.end(), x.rateTree.begin(), x.rateTree.end()); }
        detail::check(mcd_mh_set_state(mh_.get(), b.data(), d.data(), t.data(), H.data(), m.data(), v.data(), R.data(), nn));
    }
    void run(const std::vector<int32_t>& schedule, bool accumulate = false)
    {
        detail::check(mcd_mh_run(mh_.get(), schedule.data(), (int64_t)schedule.size() / stepsPerIteration_, stepsPerIteration_, accumulate, nullptr, nullptr));
    }
    void autoTune() { detail::check(mcd_mh_tune(mh_.get())); }
    I state(int64_t chain) const
    {
        const int nn = topo_.nNodes();
        Vec b(batch_), d(batch_), t(batch_), m(batch_), v(batch_), H((size_t)batch_ * nn), R((size_t)batch_ * nn);
        detail::check(mcd_mh_get_state(mh_.get(), b.data(), d.data(), t.data(), H.data(), m.data(), v.data(), R.data(), nn));
        I x;
        x.timeBirthRate = b[chain]; x.timeDeathRate = d[chain]; x.timeHeight = t[chain]; x.rateMean = m[chain]; x.rateVariance = v[chain];
        x.timeTree.assign(H.begin() + chain * nn, H.begin() + (chain + 1) * nn);
        x.rateTree.assign(R.begin() + chain * nn, R.begin() + (chain + 1) * nn);
        return x;
    }
    // [batch][3]: ln prior, ln likelihood, ln jacobianRootBranch of the current states
    Vec posterior() const
    {
        Vec post((size_t)batch_ * 3);
        detail::check(mcd_mh_get_posterior(mh_.get(), post.data()));
        return post;
    }
    int stepsPerIteration() const { return stepsPerIteration_; }
    int lastPath() const { return mcd_mh_last_path(mh_.get()); }   // MCD_MH_PATH_*
    // Metropolis-coupled MCMC: `mc3 (MC3Settings (NChains n) (SwapPeriod p) (NSwaps k))`, app/Main.hs:476-478.  mc3Init once; then
    // per period run(p iterations) + mc3Swap(k).  A host that shards the chains over GPUs passes the all-gathered ln posteriors
    // (mcd_shard_allgather of mcd_mh_posterior_device) to the three-argument form.
    void mc3Init(int nChains, const Vec& betas, uint64_t seed) { detail::check(mcd_mh_mc3_init(mh_.get(), nChains, betas.data(), batch_, seed)); }
    void mc3Swap(int nSwaps) { detail::check(mcd_mh_mc3_swap(mh_.get(), nSwaps, nullptr, 1, batch_)); }
    void mc3Swap(int nSwaps, const double* gatheredDevice, int world) { detail::check(mcd_mh_mc3_swap(mh_.get(), nSwaps, gatheredDevice, world, batch_)); }
    std::vector<int32_t> mc3Ranks() const
    {
        std::vector<int32_t> r((size_t)batch_);
        detail::check(mcd_mh_mc3_get(mh_.get(), r.data(), nullptr, nullptr, nullptr));
        return r;
    }

private:
    Topology topo_;
    std::vector<Proposal> table_;
    int64_t batch_;
    int stepsPerIteration_ = 0;
    std::unique_ptr<mcd_mh_t, detail::MhDeleter> mh_;
};

// The Hamiltonian proposal of the reference: `nutsWith calibrationsAvailable x htarget` (app/Hamiltonian.hs:95-105), with the
// position layout of `toVector` / `getMask` (:33-60) and the target prior x likelihood x jacobianRootBranch of `htargetWith`
// (:72-92).  The tree building runs on the device for all chains in lock step (csrc/k_nuts.hip), the step sizes adapt by dual
// averaging inside the library (mcd_hmc_nuts_run); the reference tunes step size and all masses (`htconf`, :62-63):
// `warmup` alternates step-size windows with pooled-variance masses.
class Nuts {
public:
    Nuts(const Likelihood& lik, const PriorFunction& prior, bool calibrationsAvailable, int64_t batch, uint64_t seed)
        : topo_(lik.topology()), batch_(batch), seed_(seed)
    {
        mcd_hmc_t* h = nullptr;
        detail::check(mcd_hmc_create(&h, lik.treeHandle(), prior.handle(), calibrationsAvailable ? 1 : 0, batch));
        h_.reset(h, [](mcd_hmc_t* p) { mcd_hmc_destroy(p); });
        dim_ = mcd_hmc_dim(h);
        eps_.assign((size_t)batch, 0.02);
        invMass_.assign((size_t)dim_, 1.0);
    }
    int dim() const { return dim_; }
    void setState(const std::vector<I>& xs)
    {
        const int nn = topo_.nNodes();
        Vec b, d, t, m, v, H, R;
        for (const I& x : xs) {
            b.push_back(x.timeBirthRate); d.push_back(x.timeDeathRate); t.push_back(x.timeHeight); m.push_back(x.rateMean); v.push_back(x.rateVariance);
            H.insert(H.end(), x.timeTree.begin(), x.timeTree.end()); R.insert(R.end(), x.rateTree.begin(), x.rateTree.end());
        }
        detail::check(mcd_hmc_set_state(h_.get(), b.data(), d.data(), t.data(), H.data(), m.data(), v.data(), R.data(), nn));
        Vec q((size_t)batch_ * dim_);
        detail::check(mcd_hmc_get_position(h_.get(), q.data(), nullptr, nullptr));
        for (int k = 0; k < dim_; ++k) {                      // first masses: (0.1 |q|)^2 averaged over the chains
            double a = 0.0;
            for (int64_t c = 0; c < batch_; ++c) a += 0.1 * std::fabs(q[(size_t)c * dim_ + k]);
            a /= (double)batch_;
            invMass_[k] = a * a > 1e-12 ? a * a : 1e-12;
        }
    }
    // step sizes by dual averaging (Hoffman & Gelman 2014, Algorithm 6), masses = pooled position variances of a window:
    // `HTuningConf HTuneLeapfrog HTuneAllMasses` (app/Hamiltonian.hs:62-63), in the library (mcd_hmc_nuts_warmup)
    void warmup(int windows = 3, int window = 60, double delta = 0.65, int maxDepth = 6)
    {
        Vec alpha((size_t)batch_);
        detail::check(mcd_hmc_nuts_warmup(h_.get(), windows, window, eps_.data(), invMass_.data(), delta, maxDepth, seed_, 0, transition_, alpha.data()));
        transition_ += (uint64_t)(windows + 1) * (uint64_t)window;
    }
    // n transitions with the tuned step sizes and masses; returns the mean acceptance statistic per chain
    Vec run(int n, int maxDepth = 6)
    {
        Vec alpha((size_t)batch_);
        detail::check(mcd_hmc_nuts_run(h_.get(), n, 0, eps_.data(), invMass_.data(), 0.65, maxDepth, seed_, 0, transition_, alpha.data(), nullptr, nullptr));
        transition_ += (uint64_t)n;
        return alpha;
    }
    I state(int64_t chain) const
    {
        const int nn = topo_.nNodes();
        Vec b(batch_), d(batch_), t(batch_), m(batch_), v(batch_), H((size_t)batch_ * nn), R((size_t)batch_ * nn);
        detail::check(mcd_hmc_get_state(h_.get(), b.data(), d.data(), t.data(), H.data(), m.data(), v.data(), R.data(), nn));
        I x;
        x.timeBirthRate = b[chain]; x.timeDeathRate = d[chain]; x.timeHeight = t[chain]; x.rateMean = m[chain]; x.rateVariance = v[chain];
        x.timeTree.assign(H.begin() + chain * nn, H.begin() + (chain + 1) * nn);
        x.rateTree.assign(R.begin() + chain * nn, R.begin() + (chain + 1) * nn);
        return x;
    }
    const Vec& stepSizes() const { return eps_; }
    const Vec& inverseMasses() const { return invMass_; }

private:
    Topology topo_;
    int64_t batch_;
    uint64_t seed_, transition_ = 0;
    int dim_ = 0;
    Vec eps_, invMass_;
    std::shared_ptr<mcd_hmc_t> h_;
};

}  // namespace mcmcdate
