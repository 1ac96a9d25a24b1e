// C++ host-mirror test (run on the GPU box by tests/test_gpu_parity.py::test_cpp_host_mirror).
// Reads a tiny text fixture written by the Python test: n_nodes, parent[], mu[], sigma_inv[], logdet,
// one state, the expected log-likelihood and log-Jacobian (from the CPU oracle); evaluates through
// mcmcdate::likelihoodFunction and checks the tolerance 1e-10 * max(1, |ll|).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../mcmc-date_amd/host/mcmcdate.hpp"

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: test_host_mirror fixture.txt\n"); return 2; }
    std::ifstream f(argv[1]);
    int nn;
    f >> nn;
    mcmcdate::Topology topo;
    topo.parent.resize(nn);
    for (auto& p : topo.parent) f >> p;
    const int n = nn - 2;
    mcmcdate::Full full;
    full.mu.resize(n);
    full.sigmaInv.resize((size_t)n * n);
    for (auto& v : full.mu) f >> v;
    for (auto& v : full.sigmaInv) f >> v;
    f >> full.logDetSigma;
    mcmcdate::I x;
    x.timeTree.resize(nn);
    x.rateTree.resize(nn);
    f >> x.timeHeight >> x.rateMean;
    for (auto& v : x.timeTree) f >> v;
    for (auto& v : x.rateTree) f >> v;
    double ll_ref, lj_ref;
    f >> ll_ref >> lj_ref;
    if (!f) { std::fprintf(stderr, "bad fixture\n"); return 2; }
    try {
        mcmcdate::Likelihood lik(mcmcdate::LikelihoodData{full}, topo);
        auto fn = mcmcdate::likelihoodFunction(mcmcdate::LikelihoodData{full}, topo);
        const double ll = fn(x), lj = lik.jacobianRootBranch(x);
        const auto d = mcmcdate::sumFirstTwo(mcmcdate::getBranches(topo, mcmcdate::heightTreeToLengthTree(topo, x.timeTree)));
        std::printf("ll=%.17g ref=%.17g lj=%.17g ref=%.17g d0=%.6g\n", ll, ll_ref, lj, lj_ref, d[0]);
        if (std::fabs(ll - ll_ref) > 1e-10 * std::fmax(1.0, std::fabs(ll_ref))) return 1;
        if (std::fabs(lj - lj_ref) > 1e-12 * std::fmax(1.0, std::fabs(lj_ref))) return 1;
        // structural fault: trifurcating root -> exception with the reference's message
        mcmcdate::Topology bad;
        bad.parent = {-1, 0, 0, 0};
        bool threw = false;
        try { mcmcdate::getBranches(bad, {0, 1, 2, 3}); } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("not bifurcating") != std::string::npos; }
        if (!threw) return 1;
        // NoData: likelihood 1.0
        if (mcmcdate::likelihoodFunction(mcmcdate::LikelihoodData{mcmcdate::NoData{}}, topo)(x) != 0.0) return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
    std::puts("ok");
    return 0;
}
