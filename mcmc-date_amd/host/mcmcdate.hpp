// mcmcdate.hpp -- C++ host-side mirror of the reference's likelihood plugin surface, on top of the
// C ABI (include/mcmcdate_mvn.h).  Header-only.  Names and argument meaning follow the reference:
//
//   LikelihoodData = Full | Sparse | Univariate | NoData            app/Probability.hs:210-235
//   likelihoodFunction :: LikelihoodData -> LikelihoodFunction I    app/Probability.hs:277-281
//   jacobianRootBranch                                              app/Probability.hs:408-410
//   getBranches / sumFirstTwo                                       app/Tools.hs:36-48
//   heightTreeToLengthTree                                          lib/Mcmc/Tree/Types.hs:224-233
//   I (state record)                                                app/State.hs:70-91
//
// The reference is Haskell; its toolchain is absent from this image, so the host layer above the
// C ABI is written in C++ (compiled code, like the reference) -- haskell/McmcDate/Gpu.hs holds the
// source-level FFI shim a maintainer would add to the reference itself.  Structural faults throw
// std::runtime_error (the reference calls `error`); numeric NaN/Inf flow through the returned value.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
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

namespace detail {
inline void check(int rc)
{
    if (rc != MCD_OK) throw std::runtime_error(std::string(mcd_last_error()));
}
struct MvnDeleter { void operator()(mcd_mvn_t* p) const { mcd_mvn_destroy(p); } };
struct TreeDeleter { void operator()(mcd_tree_t* p) const { mcd_tree_destroy(p); } };
}  // namespace detail

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

private:
    Topology topo_;
    std::unique_ptr<mcd_mvn_t, detail::MvnDeleter> mvn_;
    std::unique_ptr<mcd_tree_t, detail::TreeDeleter> tree_;
};

// likelihoodFunction :: LikelihoodData -> LikelihoodFunction I   (app/Probability.hs:277-281)
inline std::function<double(const I&)> likelihoodFunction(const LikelihoodData& lhd, const Topology& topo, int device = 0)
{
    auto lik = std::make_shared<Likelihood>(lhd, topo, device);
    return [lik](const I& x) { return (*lik)(x); };
}

}  // namespace mcmcdate
