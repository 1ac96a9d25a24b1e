// C++ host-mirror test (run on the GPU box by tests/test_gpu_parity.py::test_cpp_host_mirror).
// Reads a tiny text fixture written by the Python test: n_nodes, parent[], mu[], sigma_inv[], logdet,
// one state, the expected log-likelihood and log-Jacobian (from the CPU oracle); evaluates through
// mcmcdate::likelihoodFunction and checks the tolerance 1e-10 * max(1, |ll|).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

#include "../../mcmc-date_amd/host/mcmcdate.hpp"

// Second mode (tests/test_gpu_mh.py::test_cpp_sampler_mirror): `test_host_mirror --sampler fixture.txt`.
// Fixture: n_nodes, parent[], mu[], sigma_inv[], logdet, ht, clock model, calibrations, constraints, braces, mean branch
// lengths, batch, seed, n_sched, schedule[].  Builds priorFunction, the proposal cycle and the sampler in C++, starts all
// chains from initWith, runs the given schedule and prints the proposal table summary, the log prior of the initial state
// and the [batch][3] posterior after the run; the Python test repeats the same run through the Python mirror and expects
// identical numbers.
static int sampler_mode(const char* path)
{
    using namespace mcmcdate;
    std::ifstream f(path);
    int nn;
    f >> nn;
    Topology topo;
    topo.parent.resize(nn);
    for (auto& p : topo.parent) f >> p;
    const int n = nn - 2;
    Full full;
    full.mu.resize(n);
    full.sigmaInv.resize((size_t)n * n);
    for (auto& v : full.mu) f >> v;
    for (auto& v : full.sigmaInv) f >> v;
    f >> full.logDetSigma;
    double ht;
    int model, ncal, ncon, nbr;
    f >> ht >> model >> ncal;
    std::vector<Calibration> cb(ncal);
    for (auto& c : cb) { int hl, hh; f >> c.node >> hl >> c.lower >> c.lowerP >> hh >> c.upper >> c.upperP; c.hasLower = hl; c.hasUpper = hh; c.name = "c"; }
    f >> ncon;
    std::vector<Constraint> cs(ncon);
    for (auto& k : cs) { f >> k.young >> k.old >> k.p; k.name = "k"; }
    f >> nbr;
    std::vector<Brace> bs(nbr);
    for (auto& b : bs) { int m; f >> m >> b.sd; b.nodes.resize(m); for (auto& v : b.nodes) f >> v; b.name = "b"; }
    Vec lengths(nn);
    for (auto& v : lengths) f >> v;
    long long batch;
    unsigned long long seed;
    size_t nsched;
    f >> batch >> seed >> nsched;
    std::vector<int32_t> sched(nsched);
    for (auto& s : sched) f >> s;
    if (!f) { std::fprintf(stderr, "bad fixture\n"); return 2; }
    Likelihood lik(LikelihoodData{full}, topo);
    PriorFunction prior(ht, (RelaxedMolecularClockModel)model, cb, cs, bs, topo);
    auto ps = proposals(topo, bs, ncal > 0);
    int wsum = 0;
    long long kindsum = 0, nodesum = 0, dimsum = 0;
    for (auto& p : ps) { wsum += p.weight; kindsum += p.kind; nodesum += p.node; dimsum += p.dim; }
    std::printf("table %zu %d %lld %lld %lld %d\n", ps.size(), wsum, kindsum, nodesum, dimsum, weightNBranches(nn));
    I x0 = initWith(topo, lengths);
    if (ncal > 0) x0.timeHeight = ht;
    std::printf("prior0 %.17g lik0 %.17g\n", prior(x0), lik(x0));
    Sampler smp(lik, prior, ps, batch, seed);
    smp.setInitialState(x0);
    smp.run(sched);
    const Vec post = smp.posterior();
    for (long long b = 0; b < batch; ++b) std::printf("post %lld %.17g %.17g %.17g\n", b, post[b * 3], post[b * 3 + 1], post[b * 3 + 2]);
    const I last = smp.state(batch - 1);
    std::printf("state %.17g %.17g %.17g\n", last.timeHeight, last.timeTree[1], last.rateTree[nn - 1]);
    // the Hamiltonian proposal through the same mirror: NUTS on the device from the sampler's states, two transitions with
    // the library's random streams (seed, chain, transition)
    {
        Nuts nuts(lik, prior, ncal > 0, batch, seed);
        std::vector<I> xs;
        for (long long b = 0; b < batch; ++b) xs.push_back(smp.state(b));
        nuts.setState(xs);
        const Vec alpha = nuts.run(2, 4);
        const I xn = nuts.state(batch - 1);
        std::printf("nuts %d %.17g %.17g %.17g %.17g\n", nuts.dim(), alpha[0], alpha[batch - 1], xn.timeHeight, xn.timeTree[1]);
        // step sizes and masses tuned in the library (mcd_hmc_nuts_warmup): one window of 6 transitions + the closing window
        nuts.warmup(1, 6, 0.65, 4);
        std::printf("warmup %.17g %.17g %.17g %.17g\n", nuts.stepSizes()[0], nuts.stepSizes()[batch - 1], nuts.inverseMasses()[0],
                    nuts.inverseMasses()[nuts.dim() - 1]);
    }
    // the swap phase of Metropolis-coupled MCMC through the mirror (mcd_mh_mc3_init / mcd_mh_mc3_swap): groups of two chains
    if (batch % 2 == 0) {
        smp.mc3Init(2, Vec{1.0, 0.3}, 99);
        smp.mc3Swap(1);
        smp.mc3Swap(1);
        std::printf("mc3");
        for (int32_t r : smp.mc3Ranks()) std::printf(" %d", (int)r);
        std::printf("\n");
    }
    std::mt19937_64 rng(1);
    const auto cyc = cycleSchedule(ps, 2, rng);
    if ((int)cyc.size() != 2 * smp.stepsPerIteration()) return 1;
    std::puts("ok");
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 3 && std::string(argv[1]) == "--sampler") {
        try {
            return sampler_mode(argv[2]);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "exception: %s\n", e.what());
            return 3;
        }
    }
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
        // the same state through the matrix-core multiply form: same value to rounding, same Jacobian bits
        const mcmcdate::LogpdfForm before = mcmcdate::setLogpdfForm(mcmcdate::LogpdfForm::Multiply);   // (MCD_WIDE may preset it)
        const double ll_m = fn(x), lj_m = lik.jacobianRootBranch(x);
        if (mcmcdate::setLogpdfForm(before) != mcmcdate::LogpdfForm::Multiply) return 1;
        if (std::fabs(ll_m - ll_ref) > 1e-10 * std::fmax(1.0, std::fabs(ll_ref)) || lj_m != lj) return 1;
        // the knob table: set, read back, back to the default; unknown names are refused
        mcmcdate::setOption("MCD_MH_SEGMENTS", 0);
        if (mcmcdate::getOption("MCD_MH_SEGMENTS") != std::optional<int>(0)) return 1;
        mcmcdate::setOption("MCD_MH_SEGMENTS", std::nullopt);
        if (mcmcdate::getOption("MCD_MH_SEGMENTS").has_value()) return 1;
        try {
            mcmcdate::setOption("MCD_NO_SUCH_KNOB", 1);
            return 1;
        } catch (const std::runtime_error&) {
        }
        // structural fault: trifurcating root -> exception with the reference's message
        mcmcdate::Topology bad;
        bad.parent = {-1, 0, 0, 0};
        bool threw = false;
        try { mcmcdate::getBranches(bad, {0, 1, 2, 3}); } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("not bifurcating") != std::string::npos; }
        if (!threw) return 1;
        // NoData: likelihood 1.0
        if (mcmcdate::likelihoodFunction(mcmcdate::LikelihoodData{mcmcdate::NoData{}}, topo)(x) != 0.0) return 1;
        // the same precision matrix as an association list through the sparse form (mcd_sparse_*): the same value to rounding
        mcmcdate::Sparse sp;
        sp.mu = full.mu;
        sp.logDetSigma = full.logDetSigma;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (full.sigmaInv[(size_t)i * n + j] != 0.0) sp.sigmaInvAssoc.push_back({{i, j}, full.sigmaInv[(size_t)i * n + j]});
        mcmcdate::SparseLikelihood slik(sp, topo);
        const double ll_s = slik(x);
        std::printf("ll_sparse=%.17g\n", ll_s);
        if (std::fabs(ll_s - ll_ref) > 1e-10 * std::fmax(1.0, std::fabs(ll_ref))) return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
    std::puts("ok");
    return 0;
}
