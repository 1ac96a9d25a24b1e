// hmc_capi.cpp -- C ABI of the device leapfrog (include/mcmcdate_mvn.h, "mcd_hmc_*").  The state of `batch` chains, their
// momenta and gradients stay on the device; one leapfrog step = kick, drift (k_hmc.hip), prior gradient (k_prior_grad.hip),
// likelihood gradient (k_tree_grad.hip).  No CPU path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <memory>
#include <vector>

#include "../../include/mcmcdate_mvn.h"
#include "mvn_kernels.h"

extern "C" int mcd_set_last_error_(int code, const char* msg);   // mvn_capi.cpp

namespace {

int hfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return mcd_set_last_error_(code, buf);
}

#define HHIP_TRY(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return hfail(MCD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

}  // namespace

struct mcd_hmc {
    int device = 0;
    const mcd::MvnDev* mvn = nullptr;
    const mcd::TreeDev* tree = nullptr;
    const mcd::PriorDev* prior = nullptr;
    mcd::HmcDev dev{};
    double *d_eps = nullptr, *d_dir = nullptr, *d_inv_mass = nullptr;
    mcd::NutsDev nuts{};           // allocated by the first NUTS transition (mcd_hmc_nuts)
    int* d_active = nullptr;
    bool have_state = false;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;

    ~mcd_hmc()
    {
        (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

template <class T>
int halloc(mcd_hmc* m, T** p, size_t count)
{
    *p = nullptr;
    HHIP_TRY(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
    m->allocs.push_back(*p);
    HHIP_TRY(hipMemset(*p, 0, sizeof(T) * (count ? count : 1)));
    return MCD_OK;
}

// ln target and its gradient at the current state: two batched gradient launches
int eval_gradients(mcd_hmc* m)
{
    const mcd::HmcDev& D = m->dev;
    const int64_t B = D.batch;
    HHIP_TRY(mcd::launch_prior_grad(*m->prior, D.sc, D.sc + B, D.sc + 2 * B, D.H, D.sc + 3 * B, D.sc + 4 * B, D.R, D.ld, B, D.lp, D.gp_sc,
                                    D.gp_sc + B, D.gp_sc + 2 * B, D.gp_H, D.gp_sc + 3 * B, D.gp_sc + 4 * B, D.gp_R, m->stream));
    HHIP_TRY(mcd::launch_tree_grad(*m->mvn, *m->tree, D.H, D.R, D.ld, D.sc + 2 * B, D.sc + 3 * B, B, D.ll, D.gl_H, D.gl_R, D.gl_tH, D.gl_rMu,
                                   m->stream));
    return MCD_OK;
}

}  // namespace

extern "C" {

int mcd_hmc_create(mcd_hmc_t** out, const mcd_tree_t* tree, const mcd_prior_t* prior, int calibrations_available, int64_t batch)
{
    if (!out) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_create: out is NULL");
    *out = nullptr;
    if (!tree || !prior) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_create: NULL tree or prior handle");
    if (batch <= 0) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_create: batch must be positive");
    std::unique_ptr<mcd_hmc> m(new mcd_hmc());
    int dev_t = 0, dev_p = 0;
    const int32_t* parent = nullptr;
    const double* host_L = nullptr;
    if (mcd_tree_internal_(tree, &m->mvn, &m->tree, &dev_t, &parent, &host_L) || mcd_prior_internal_(prior, &m->prior, &dev_p))
        return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_create: invalid handle");
    if (dev_t != dev_p) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_create: tree and prior live on different GPUs");
    const int n = m->tree->n_nodes;
    if (m->prior->n_nodes != n) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_create: tree has %d nodes, prior %d", n, m->prior->n_nodes);
    // getMask (app/Hamiltonian.hs:33-47) in fold order of the state record, then toVector's reverse order (:49-53)
    std::vector<char> leaf(n, 1);
    for (int v = 1; v < n; ++v) leaf[parent[v]] = 0;
    std::vector<int32_t> field, index;
    auto push = [&](int f, int v) {
        field.push_back(f);
        index.push_back(v);
    };
    push(0, 0);
    push(1, 0);
    if (calibrations_available) push(2, 0);
    for (int v = 1; v < n; ++v)
        if (!leaf[v]) push(3, v);                 // root height and leaf heights are masked
    push(4, 0);
    push(5, 0);
    for (int v = 1; v < n; ++v) push(6, v);       // the stem of the rate tree is masked
    std::vector<int32_t> rf(field.rbegin(), field.rend()), ri(index.rbegin(), index.rend());
    const int dim = (int)rf.size();
    m->device = dev_t;
    HHIP_TRY(hipSetDevice(m->device));
    HHIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    mcd::HmcDev& D = m->dev;
    D.n_nodes = n;
    D.dim = dim;
    D.root_right = m->tree->root_right;
    D.batch = batch;
    D.ld = (n + 7) / 8 * 8;
    const size_t B = (size_t)batch, BL = B * (size_t)D.ld, BD = B * (size_t)dim;
    int32_t *pf = nullptr, *pi = nullptr;
    int rc = MCD_OK;
    if ((rc = halloc(m.get(), &pf, (size_t)dim)) || (rc = halloc(m.get(), &pi, (size_t)dim)) || (rc = halloc(m.get(), &D.sc, 5 * B)) ||
        (rc = halloc(m.get(), &D.H, BL)) || (rc = halloc(m.get(), &D.R, BL)) || (rc = halloc(m.get(), &D.lp, B)) ||
        (rc = halloc(m.get(), &D.gp_sc, 5 * B)) || (rc = halloc(m.get(), &D.gp_H, BL)) || (rc = halloc(m.get(), &D.gp_R, BL)) ||
        (rc = halloc(m.get(), &D.ll, B)) || (rc = halloc(m.get(), &D.gl_H, BL)) || (rc = halloc(m.get(), &D.gl_R, BL)) ||
        (rc = halloc(m.get(), &D.gl_tH, B)) || (rc = halloc(m.get(), &D.gl_rMu, B)) || (rc = halloc(m.get(), &D.q, BD)) ||
        (rc = halloc(m.get(), &D.p, BD)) || (rc = halloc(m.get(), &D.grad, BD)) || (rc = halloc(m.get(), &D.value, B)) ||
        (rc = halloc(m.get(), &m->d_eps, B)) || (rc = halloc(m.get(), &m->d_dir, B)) || (rc = halloc(m.get(), &m->d_inv_mass, (size_t)dim)))
        return rc;
    HHIP_TRY(hipMemcpy(pf, rf.data(), sizeof(int32_t) * dim, hipMemcpyHostToDevice));
    HHIP_TRY(hipMemcpy(pi, ri.data(), sizeof(int32_t) * dim, hipMemcpyHostToDevice));
    D.pos_field = pf;
    D.pos_index = pi;
    D.eps = m->d_eps;
    D.dir = m->d_dir;
    D.inv_mass = m->d_inv_mass;
    *out = m.release();
    return MCD_OK;
}

void mcd_hmc_destroy(mcd_hmc_t* m) { delete m; }

int mcd_hmc_dim(const mcd_hmc_t* m) { return m ? m->dev.dim : hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_dim: NULL handle"); }

int mcd_hmc_set_state(mcd_hmc_t* m, const double* birth, const double* death, const double* tH, const double* heights,
                      const double* rMu, const double* rVar, const double* rates, int64_t ld_state)
{
    if (!m || !birth || !death || !tH || !heights || !rMu || !rVar || !rates) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_set_state: NULL argument");
    mcd::HmcDev& D = m->dev;
    if (ld_state < D.n_nodes) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_set_state: ld_state < n_nodes");
    HHIP_TRY(hipSetDevice(m->device));
    const size_t B = (size_t)D.batch;
    const double* src[5] = {birth, death, tH, rMu, rVar};
    for (int i = 0; i < 5; ++i) HHIP_TRY(hipMemcpyAsync(D.sc + i * B, src[i], sizeof(double) * B, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpy2DAsync(D.H, sizeof(double) * D.ld, heights, sizeof(double) * ld_state, sizeof(double) * D.n_nodes, B, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpy2DAsync(D.R, sizeof(double) * D.ld, rates, sizeof(double) * ld_state, sizeof(double) * D.n_nodes, B, hipMemcpyHostToDevice, m->stream));
    if (int rc = eval_gradients(m)) return rc;
    HHIP_TRY(mcd::launch_hmc_collect(D, m->stream));
    HHIP_TRY(hipStreamSynchronize(m->stream));
    m->have_state = true;
    return MCD_OK;
}

int mcd_hmc_get_state(const mcd_hmc_t* cm, double* birth, double* death, double* tH, double* heights, double* rMu, double* rVar,
                      double* rates, int64_t ld_state)
{
    if (!cm || !birth || !death || !tH || !heights || !rMu || !rVar || !rates) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_get_state: NULL argument");
    const mcd::HmcDev& D = cm->dev;
    if (ld_state < D.n_nodes) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_get_state: ld_state < n_nodes");
    HHIP_TRY(hipSetDevice(cm->device));
    HHIP_TRY(hipStreamSynchronize(cm->stream));
    const size_t B = (size_t)D.batch;
    double* dst[5] = {birth, death, tH, rMu, rVar};
    for (int i = 0; i < 5; ++i) HHIP_TRY(hipMemcpy(dst[i], D.sc + i * B, sizeof(double) * B, hipMemcpyDeviceToHost));
    HHIP_TRY(hipMemcpy2D(heights, sizeof(double) * ld_state, D.H, sizeof(double) * D.ld, sizeof(double) * D.n_nodes, B, hipMemcpyDeviceToHost));
    HHIP_TRY(hipMemcpy2D(rates, sizeof(double) * ld_state, D.R, sizeof(double) * D.ld, sizeof(double) * D.n_nodes, B, hipMemcpyDeviceToHost));
    return MCD_OK;
}

int mcd_hmc_get_position(const mcd_hmc_t* cm, double* q, double* value, double* grad)
{
    if (!cm) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_get_position: NULL handle");
    if (!cm->have_state) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_get_position: call mcd_hmc_set_state first");
    const mcd::HmcDev& D = cm->dev;
    HHIP_TRY(hipSetDevice(cm->device));
    HHIP_TRY(hipStreamSynchronize(cm->stream));
    const size_t B = (size_t)D.batch, BD = B * (size_t)D.dim;
    if (q) HHIP_TRY(hipMemcpy(q, D.q, sizeof(double) * BD, hipMemcpyDeviceToHost));
    if (value) HHIP_TRY(hipMemcpy(value, D.value, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (grad) HHIP_TRY(hipMemcpy(grad, D.grad, sizeof(double) * BD, hipMemcpyDeviceToHost));
    return MCD_OK;
}

int mcd_hmc_leapfrog(mcd_hmc_t* m, double* p, const double* eps, const double* dir, const double* inv_mass, int n_steps)
{
    if (!m || !p || !eps || !inv_mass) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_leapfrog: NULL argument");
    if (!m->have_state) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_leapfrog: call mcd_hmc_set_state first");
    if (n_steps < 0) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_leapfrog: negative number of steps");
    mcd::HmcDev& D = m->dev;
    const size_t B = (size_t)D.batch, BD = B * (size_t)D.dim;
    for (int k = 0; k < D.dim; ++k)
        if (!(inv_mass[k] > 0) || !std::isfinite(inv_mass[k])) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_leapfrog: inverse masses must be positive");
    HHIP_TRY(hipSetDevice(m->device));
    HHIP_TRY(hipMemcpyAsync(D.p, p, sizeof(double) * BD, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpyAsync(m->d_eps, eps, sizeof(double) * B, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpyAsync(m->d_inv_mass, inv_mass, sizeof(double) * D.dim, hipMemcpyHostToDevice, m->stream));
    if (dir) {
        HHIP_TRY(hipMemcpyAsync(m->d_dir, dir, sizeof(double) * B, hipMemcpyHostToDevice, m->stream));
        D.dir = m->d_dir;
    } else {
        D.dir = nullptr;
    }
    // p += eps/2 g;  [ q += eps Minv p;  g = grad(q);  p += eps g ] x (n - 1);  q += eps Minv p;  g = grad(q);  p += eps/2 g
    // three launches per step (kick + drift fused, prior gradient, likelihood gradient), the closing half kick with the collect
    for (int s = 0; s < n_steps; ++s) {
        HHIP_TRY(mcd::launch_hmc_kick_drift(D, (s == 0) ? 0.5 : 1.0, m->stream));
        if (int rc = eval_gradients(m)) return rc;
    }
    if (n_steps > 0)
        HHIP_TRY(mcd::launch_hmc_kick_collect(D, 0.5, m->stream));
    else
        HHIP_TRY(mcd::launch_hmc_collect(D, m->stream));
    HHIP_TRY(hipMemcpyAsync(p, D.p, sizeof(double) * BD, hipMemcpyDeviceToHost, m->stream));
    HHIP_TRY(hipStreamSynchronize(m->stream));
    return MCD_OK;
}

int mcd_hmc_step_from(mcd_hmc_t* m, double* q, double* p, double* grad, int have_grad, const double* eps, const double* dir,
                      const double* inv_mass, double* value)
{
    if (!m || !q || !p || !grad || !eps || !inv_mass) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_step_from: NULL argument");
    if (!m->have_state) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_step_from: call mcd_hmc_set_state first (fixes the masked entries)");
    mcd::HmcDev& D = m->dev;
    const size_t B = (size_t)D.batch, BD = B * (size_t)D.dim;
    for (int k = 0; k < D.dim; ++k)
        if (!(inv_mass[k] > 0) || !std::isfinite(inv_mass[k])) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_step_from: inverse masses must be positive");
    HHIP_TRY(hipSetDevice(m->device));
    HHIP_TRY(hipMemcpyAsync(D.q, q, sizeof(double) * BD, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpyAsync(D.p, p, sizeof(double) * BD, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpyAsync(m->d_eps, eps, sizeof(double) * B, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpyAsync(m->d_inv_mass, inv_mass, sizeof(double) * D.dim, hipMemcpyHostToDevice, m->stream));
    if (dir) {
        HHIP_TRY(hipMemcpyAsync(m->d_dir, dir, sizeof(double) * B, hipMemcpyHostToDevice, m->stream));
        D.dir = m->d_dir;
    } else {
        D.dir = nullptr;
    }
    HHIP_TRY(mcd::launch_hmc_scatter(D, m->stream));
    if (have_grad) {
        HHIP_TRY(hipMemcpyAsync(D.grad, grad, sizeof(double) * BD, hipMemcpyHostToDevice, m->stream));
    } else {
        if (int rc = eval_gradients(m)) return rc;
    }
    HHIP_TRY(mcd::launch_hmc_kick(D, 0.5, have_grad ? 1 : 0, m->stream));
    HHIP_TRY(mcd::launch_hmc_drift(D, m->stream));
    if (int rc = eval_gradients(m)) return rc;
    HHIP_TRY(mcd::launch_hmc_kick(D, 0.5, 0, m->stream));
    HHIP_TRY(mcd::launch_hmc_collect(D, m->stream));
    HHIP_TRY(hipMemcpyAsync(q, D.q, sizeof(double) * BD, hipMemcpyDeviceToHost, m->stream));
    HHIP_TRY(hipMemcpyAsync(p, D.p, sizeof(double) * BD, hipMemcpyDeviceToHost, m->stream));
    HHIP_TRY(hipMemcpyAsync(grad, D.grad, sizeof(double) * BD, hipMemcpyDeviceToHost, m->stream));
    if (value) HHIP_TRY(hipMemcpyAsync(value, D.value, sizeof(double) * B, hipMemcpyDeviceToHost, m->stream));
    HHIP_TRY(hipStreamSynchronize(m->stream));
    return MCD_OK;
}

}  // extern "C"

// ---- NUTS (k_nuts.hip) ------------------------------------------------------------------------------------------------
namespace {

constexpr int kNutsMaxDepth = 12;

int nuts_alloc(mcd_hmc* m)
{
    if (m->nuts.qm) return MCD_OK;
    const mcd::HmcDev& D = m->dev;
    const size_t B = (size_t)D.batch, BD = B * (size_t)D.dim;
    mcd::NutsDev& N = m->nuts;
    N.max_depth = kNutsMaxDepth;
    int rc = MCD_OK;
    double** pd[] = {&N.qm, &N.pm, &N.gm, &N.qp, &N.pp, &N.gp, &N.qc, &N.gc, &N.qn, &N.gn};
    for (double** p : pd)
        if ((rc = halloc(m, p, BD))) return rc;
    if ((rc = halloc(m, &N.sq, BD * kNutsMaxDepth)) || (rc = halloc(m, &N.sp, BD * kNutsMaxDepth))) return rc;
    double** pb[] = {&N.lpc, &N.lpn, &N.log_u, &N.joint0, &N.alpha};
    for (double** p : pb)
        if ((rc = halloc(m, p, B))) return rc;
    int** pi[] = {&N.j, &N.v, &N.i, &N.n, &N.n1, &N.s1, &N.done, &N.n_alpha, &N.depth, &N.leaf};
    for (int** p : pi)
        if ((rc = halloc(m, p, B))) return rc;
    return halloc(m, &m->d_active, 1);
}

// one transition for every chain; eps and inv_mass already on the device
int nuts_transition(mcd_hmc* m, int max_depth, uint64_t seed, int64_t chain0, uint64_t transition)
{
    mcd::HmcDev& D = m->dev;
    D.dir = m->d_dir;
    HHIP_TRY(mcd::launch_nuts_begin(D, m->nuts, seed, chain0, transition, m->stream));
    const int64_t max_rounds = (int64_t)1 << max_depth;               // 2^max_depth - 1 leaves at most
    for (int64_t r = 0; r < max_rounds; ++r) {
        if (int rc = eval_gradients(m)) return rc;
        HHIP_TRY(hipMemsetAsync(m->d_active, 0, sizeof(int), m->stream));
        HHIP_TRY(mcd::launch_nuts_step(D, m->nuts, seed, chain0, transition, max_depth, m->d_active, m->stream));
        int active = 0;
        HHIP_TRY(hipMemcpyAsync(&active, m->d_active, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HHIP_TRY(hipStreamSynchronize(m->stream));
        if (active == 0) break;
    }
    HHIP_TRY(mcd::launch_nuts_end(D, m->nuts, m->stream));
    // k_nuts_end put the accepted point into D.q / D.grad / D.value and the state arrays, but the raw outputs of the gradient
    // kernels still belong to the LAST LEAF evaluated (another point, possibly outside the support).  mcd_hmc_leapfrog's first
    // half kick reads those raw outputs: evaluate them at the accepted point so that every entry point may follow a transition.
    if (int rc = eval_gradients(m)) return rc;
    HHIP_TRY(mcd::launch_hmc_collect(D, m->stream));
    return MCD_OK;
}

}  // namespace

extern "C" {

int mcd_hmc_nuts(mcd_hmc_t* m, const double* eps, const double* inv_mass, int max_depth, uint64_t seed, int64_t chain_offset,
                 uint64_t transition, double* alpha, int32_t* depth)
{
    if (!m || !eps || !inv_mass) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts: NULL argument");
    if (!m->have_state) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts: call mcd_hmc_set_state first");
    if (max_depth < 1 || max_depth > kNutsMaxDepth) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts: max_depth must be 1 .. %d", kNutsMaxDepth);
    mcd::HmcDev& D = m->dev;
    const size_t B = (size_t)D.batch;
    for (int k = 0; k < D.dim; ++k)
        if (!(inv_mass[k] > 0) || !std::isfinite(inv_mass[k])) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts: inverse masses must be positive");
    for (size_t b = 0; b < B; ++b)
        if (!(eps[b] > 0) || !std::isfinite(eps[b])) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts: step sizes must be positive");
    HHIP_TRY(hipSetDevice(m->device));
    if (int rc = nuts_alloc(m)) return rc;
    HHIP_TRY(hipMemcpyAsync(m->d_eps, eps, sizeof(double) * B, hipMemcpyHostToDevice, m->stream));
    HHIP_TRY(hipMemcpyAsync(m->d_inv_mass, inv_mass, sizeof(double) * D.dim, hipMemcpyHostToDevice, m->stream));
    if (int rc = nuts_transition(m, max_depth, seed, chain_offset, transition)) return rc;
    std::vector<double> a(B);
    std::vector<int> na(B), dp(B);
    HHIP_TRY(hipMemcpyAsync(a.data(), m->nuts.alpha, sizeof(double) * B, hipMemcpyDeviceToHost, m->stream));
    HHIP_TRY(hipMemcpyAsync(na.data(), m->nuts.n_alpha, sizeof(int) * B, hipMemcpyDeviceToHost, m->stream));
    HHIP_TRY(hipMemcpyAsync(dp.data(), m->nuts.depth, sizeof(int) * B, hipMemcpyDeviceToHost, m->stream));
    HHIP_TRY(hipStreamSynchronize(m->stream));
    for (size_t b = 0; b < B; ++b) {
        if (alpha) alpha[b] = a[b] / (double)(na[b] > 0 ? na[b] : 1);
        if (depth) depth[b] = dp[b];
    }
    return MCD_OK;
}

int mcd_hmc_nuts_run(mcd_hmc_t* m, int n_transitions, int adapt, double* eps, const double* inv_mass, double delta, int max_depth,
                     uint64_t seed, int64_t chain_offset, uint64_t first_transition, double* mean_alpha, double* q_mean, double* q_var)
{
    if (!m || !eps || !inv_mass) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts_run: NULL argument");
    if (n_transitions < 0) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts_run: negative number of transitions");
    if (adapt && !(delta > 0 && delta < 1)) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts_run: delta must be in (0, 1)");
    const mcd::HmcDev& D = m->dev;
    const size_t B = (size_t)D.batch, dim = (size_t)D.dim;
    // dual averaging of ln eps per chain: Hoffman & Gelman (2014), Algorithm 6 (gamma = 0.05, t0 = 10, kappa = 0.75)
    const double gamma = 0.05, t0 = 10.0, kappa = 0.75;
    std::vector<double> mu(B), h_bar(B, 0.0), log_eps_bar(B, 0.0), cur(eps, eps + B), alpha(B), asum(B, 0.0);
    for (size_t b = 0; b < B; ++b) mu[b] = std::log(10.0 * eps[b]);
    std::vector<double> q(q_mean || q_var ? B * dim : 0), s1(dim, 0.0), s2(dim, 0.0);
    for (int t = 1; t <= n_transitions; ++t) {
        if (int rc = mcd_hmc_nuts(m, cur.data(), inv_mass, max_depth, seed, chain_offset, first_transition + (uint64_t)(t - 1), alpha.data(), nullptr))
            return rc;
        for (size_t b = 0; b < B; ++b) {
            asum[b] += alpha[b];
            if (adapt) {
                h_bar[b] = (1.0 - 1.0 / (t + t0)) * h_bar[b] + (delta - alpha[b]) / (t + t0);
                const double log_eps = mu[b] - std::sqrt((double)t) / gamma * h_bar[b];
                const double w = std::pow((double)t, -kappa);
                log_eps_bar[b] = w * log_eps + (1.0 - w) * log_eps_bar[b];
                cur[b] = std::exp(log_eps);
            }
        }
        if (!q.empty()) {
            HHIP_TRY(hipMemcpy(q.data(), D.q, sizeof(double) * B * dim, hipMemcpyDeviceToHost));
            for (size_t b = 0; b < B; ++b)
                for (size_t k = 0; k < dim; ++k) {
                    const double x = q[b * dim + k];
                    s1[k] += x;
                    s2[k] += x * x;
                }
        }
    }
    for (size_t b = 0; b < B; ++b) {
        if (adapt && n_transitions > 0) eps[b] = std::exp(log_eps_bar[b]);
        if (mean_alpha) mean_alpha[b] = n_transitions > 0 ? asum[b] / n_transitions : 0.0;
    }
    const double cnt = (double)B * (double)(n_transitions > 0 ? n_transitions : 1);
    for (size_t k = 0; k < dim && !q.empty(); ++k) {
        const double mean = s1[k] / cnt;
        if (q_mean) q_mean[k] = mean;
        if (q_var) q_var[k] = s2[k] / cnt - mean * mean;          // pooled over chains and transitions (what mass tuning uses)
    }
    return MCD_OK;
}

// Step sizes AND masses: the reference tunes both (`HTuningConf HTuneLeapfrog HTuneAllMasses`, app/Hamiltonian.hs:62-63; the
// schedule of the package `mcmc` is not restated).  `windows` windows of `window` transitions: dual averaging of the step sizes
// inside a window (mcd_hmc_nuts_run, adapt = 1), then the inverse masses become the pooled variance of the positions the
// window visited, shrunk towards 1e-3 (Stan's regularisation); a closing window adapts the step sizes to the final masses.
int mcd_hmc_nuts_warmup(mcd_hmc_t* m, int windows, int window, double* eps, double* inv_mass, double delta, int max_depth, uint64_t seed,
                        int64_t chain_offset, uint64_t first_transition, double* mean_alpha)
{
    if (!m || !eps || !inv_mass) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts_warmup: NULL argument");
    if (windows < 0 || window < 1) return hfail(MCD_ERR_INVALID_ARG, "mcd_hmc_nuts_warmup: need windows >= 0 and window >= 1");
    const size_t B = (size_t)m->dev.batch, dim = (size_t)m->dev.dim;
    std::vector<double> qv(dim), alpha(B);
    uint64_t t = first_transition;
    for (int w = 0; w < windows; ++w) {
        if (int rc = mcd_hmc_nuts_run(m, window, 1, eps, inv_mass, delta, max_depth, seed, chain_offset, t, alpha.data(), nullptr, qv.data())) return rc;
        t += (uint64_t)window;
        const double n_eff = (double)B * (double)window;
        for (size_t k = 0; k < dim; ++k) {
            const double v = (n_eff / (n_eff + 5.0)) * qv[k] + 1e-3 * (5.0 / (n_eff + 5.0));
            inv_mass[k] = (v > 0 && std::isfinite(v)) ? v : inv_mass[k];      // a chain outside the support leaves NaN positions: keep the mass
        }
    }
    if (int rc = mcd_hmc_nuts_run(m, window, 1, eps, inv_mass, delta, max_depth, seed, chain_offset, t, alpha.data(), nullptr, nullptr)) return rc;
    if (mean_alpha) std::copy(alpha.begin(), alpha.end(), mean_alpha);
    return MCD_OK;
}

}  // extern "C"
