// mvn_capi.cpp -- implementation of include/mcmcdate_mvn.h on top of the HIP kernels.
//
// There is no CPU fallback in this library: every evaluation entry point runs the HIP kernels
// of mvn_kernels.hip or fails with MCD_ERR_NO_DEVICE / MCD_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mcmcdate_mvn.h"
#include "host_factor.h"
#include "mvn_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(MCD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// Per-call scratch for the host-pointer entry points: a stream plus grow-only device buffers.
// Pooled per handle so that concurrent callers never share one.
struct Workspace {
    hipStream_t stream = nullptr;
    double* dbuf = nullptr;
    size_t cap = 0;  // doubles
    // small calls (the reference's one-state-per-call pattern): a pinned, device-mapped host buffer the kernel reads and
    // writes directly -- no staging copies, one launch + one synchronisation per call
    double* hbuf = nullptr;      // host address
    double* hbuf_dev = nullptr;  // the same memory as the device sees it
    size_t hcap = 0;             // doubles
};

struct WorkspacePool {
    std::mutex mu;
    std::vector<Workspace*> idle;

    Workspace* acquire()
    {
        {
            std::lock_guard<std::mutex> g(mu);
            if (!idle.empty()) {
                Workspace* w = idle.back();
                idle.pop_back();
                return w;
            }
        }
        Workspace* w = new Workspace();
        if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
            delete w;
            return nullptr;
        }
        return w;
    }
    void release(Workspace* w)
    {
        std::lock_guard<std::mutex> g(mu);
        idle.push_back(w);
    }
    void clear()
    {
        std::lock_guard<std::mutex> g(mu);
        for (Workspace* w : idle) {
            if (w->dbuf) (void)hipFree(w->dbuf);
            if (w->hbuf) (void)hipHostFree(w->hbuf);
            if (w->stream) (void)hipStreamDestroy(w->stream);
            delete w;
        }
        idle.clear();
    }
};

int ensure(Workspace* w, size_t doubles)
{
    if (w->cap >= doubles) return MCD_OK;
    if (w->dbuf) HIP_TRY(hipFree(w->dbuf));
    w->dbuf = nullptr;
    w->cap = 0;
    size_t want = doubles + doubles / 4 + 1024;
    HIP_TRY(hipMalloc((void**)&w->dbuf, want * sizeof(double)));
    w->cap = want;
    return MCD_OK;
}

constexpr size_t kZeroCopyDoubles = 8192;   // calls moving at most 64 KiB use the mapped host buffer

int ensure_mapped(Workspace* w, size_t doubles)
{
    if (w->hcap >= doubles) return MCD_OK;
    if (w->hbuf) HIP_TRY(hipHostFree(w->hbuf));
    w->hbuf = nullptr;
    w->hbuf_dev = nullptr;
    w->hcap = 0;
    HIP_TRY(hipHostMalloc((void**)&w->hbuf, kZeroCopyDoubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostGetDevicePointer((void**)&w->hbuf_dev, w->hbuf, 0));
    w->hcap = kZeroCopyDoubles;
    return doubles <= w->hcap ? MCD_OK : MCD_ERR_INVALID_ARG;
}

struct WsGuard {
    WorkspacePool* pool;
    Workspace* w;
    ~WsGuard()
    {
        if (w) pool->release(w);
    }
};

}  // namespace

struct mcd_mvn {
    int n = 0, R = 0, device = 0;
    double logdet = 0.0;
    mcd::MvnDev dev{};
    double *d_mu = nullptr, *d_invdiag = nullptr, *d_Ft = nullptr, *d_Ut = nullptr, *d_Wt = nullptr, *d_Wtb = nullptr, *d_Wc = nullptr;
    std::vector<double> L;  // host copy of the factor (row-major lower)
    mutable WorkspacePool pool;
    mcd::SplitHost* split = nullptr;   // k_split.hip: schedules of the row-split form + scratch sets per stream
    mutable int form = 0;              // MCD_FORM_* chosen for this handle (0: the process default); read with atomic loads

    ~mcd_mvn()
    {
        (void)hipSetDevice(device);
        pool.clear();
        if (d_mu) (void)hipFree(d_mu);
        if (d_invdiag) (void)hipFree(d_invdiag);
        if (d_Ft) (void)hipFree(d_Ft);
        if (d_Ut) (void)hipFree(d_Ut);
        if (d_Wt) (void)hipFree(d_Wt);
        mcd::split_host_destroy(split);
        if (d_Wtb) (void)hipFree(d_Wtb);
        if (d_Wc) (void)hipFree(d_Wc);
    }
};

struct mcd_tree {
    const mcd_mvn* mvn = nullptr;
    int n_nodes = 0;
    mcd::TreeDev dev{};
    int32_t *d_parent = nullptr, *d_slot = nullptr, *d_cptr = nullptr, *d_cidx = nullptr;
    std::vector<int32_t> parent;   // host copy

    ~mcd_tree()
    {
        if (mvn) (void)hipSetDevice(mvn->device);
        if (d_parent) (void)hipFree(d_parent);
        if (d_slot) (void)hipFree(d_slot);
        if (d_cptr) (void)hipFree(d_cptr);
        if (d_cidx) (void)hipFree(d_cidx);
    }
};

// shared with prior_capi.cpp: set this thread's error message, return the code
extern "C" int mcd_set_last_error_(int code, const char* msg)
{
    g_last_error = msg ? msg : "";
    return code;
}

int mcd_tree_internal_(const mcd_tree* t, const mcd::MvnDev** mvn, const mcd::TreeDev** tree, int* device, const int32_t** host_parent,
                       const double** host_L)
{
    if (!t || !t->mvn) return MCD_ERR_INVALID_ARG;
    *mvn = &t->mvn->dev;
    *tree = &t->dev;
    *device = t->mvn->device;
    *host_parent = t->parent.data();
    *host_L = t->mvn->L.data();
    return MCD_OK;
}

extern "C" {

const char* mcd_version(void) { return "mcmc-date_amd 0.1 (gfx950, column-sweep TRSV)"; }

const char* mcd_last_error(void) { return g_last_error.c_str(); }

int mcd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mcd_mvn_create(mcd_mvn_t** out, int n, const double* mu, const double* mat, int mat_kind, double logdet_sigma,
                   int device_id)
{
    if (!out) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_create: out is NULL");
    *out = nullptr;
    if (n < 1 || !mu || !mat) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_create: need n >= 1, mu and mat");
    if (n > MCD_MAX_DIM)
        return fail(MCD_ERR_UNSUPPORTED, "mcd_mvn_create: n = %d exceeds MCD_MAX_DIM = %d", n, MCD_MAX_DIM);
    if (mat_kind != MCD_MAT_SIGMA && mat_kind != MCD_MAT_SIGMA_INV)
        return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_create: mat_kind must be MCD_MAT_SIGMA or MCD_MAT_SIGMA_INV");
    const int ndev = mcd_device_count();
    if (ndev <= 0) return fail(MCD_ERR_NO_DEVICE, "mcd_mvn_create: no HIP device available (this library has no CPU path)");
    if (device_id < 0 || device_id >= ndev)
        return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_create: device_id %d out of range [0, %d)", device_id, ndev);
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(mu[i])) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_create: mu[%d] is not finite", i);

    // symmetric part (the quadratic form only sees it; the reference trusts symmetry, app/Main.hs:93)
    std::vector<double> A((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double a = mat[(size_t)i * n + j], b = mat[(size_t)j * n + i];
            if (!std::isfinite(a)) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_create: mat[%d][%d] is not finite", i, j);
            A[(size_t)i * n + j] = 0.5 * (a + b);
        }
    std::unique_ptr<mcd_mvn> h(new mcd_mvn());
    std::vector<double> W;                               // W = L^-1: the factor of the multiply forms (W^T W = Sigma^-1)
    if (mat_kind == MCD_MAT_SIGMA_INV) {
        // the reference's own operand (app/Main.hs:240: .data files carry Sigma^-1 and log det Sigma): factored directly,
        // nothing is inverted twice.  A precision matrix that is not positive definite is refused (the reference would
        // evaluate its quadratic form all the same; `prepare` never writes one: it aborts unless det Sigma > 0, :231).
        if (!std::isfinite(logdet_sigma)) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_create: logdet_sigma is not finite");
        if (!mcd::precision_factors(n, A, W, h->L))
            return fail(MCD_ERR_NOT_SPD, "mcd_mvn_create: inverse covariance matrix is not positive definite");
    } else {
        if (!mcd::cholesky_lower(n, A, h->L))
            return fail(MCD_ERR_NOT_SPD, "mcd_mvn_create: covariance matrix is not positive definite");
        mcd::invert_factor(n, h->L, W);
    }
    h->n = n;
    h->R = mcd::padded_blocks(n);
    h->device = device_id;
    if (h->R < 0) return fail(MCD_ERR_UNSUPPORTED, "mcd_mvn_create: unsupported dimension %d", n);
    if (mat_kind == MCD_MAT_SIGMA_INV) {
        h->logdet = logdet_sigma;
    } else {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += std::log(h->L[(size_t)i * n + i]);
        h->logdet = 2.0 * s;
    }
    std::vector<double> mu_pad, invdiag, Ft, Ut;
    mcd::pack_factors(n, h->R, h->L, mu_pad, mu, invdiag, Ft, Ut);

    HIP_TRY(hipSetDevice(device_id));
    const size_t NP = (size_t)64 * h->R;
    HIP_TRY(hipMalloc((void**)&h->d_mu, NP * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_invdiag, NP * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_Ft, NP * NP * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->d_Ut, NP * NP * sizeof(double)));
    HIP_TRY(hipMemcpy(h->d_mu, mu_pad.data(), NP * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_invdiag, invdiag.data(), NP * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_Ft, Ft.data(), NP * NP * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_Ut, Ut.data(), NP * NP * sizeof(double), hipMemcpyHostToDevice));

    const int ch = mcd::sweep_chunk_columns(h->R);
    h->dev.n = n;
    h->dev.R = h->R;
    h->dev.ncols = ((n + ch - 1) / ch) * ch;
    h->dev.c = -(0.9189385332046727417803297364056176 * (double)n);  // m_ln_sqrt_2_pi * k, Probability.hs:172-173
    h->dev.logdet = h->logdet;
    h->dev.mu = h->d_mu;
    h->dev.invdiag = h->d_invdiag;
    h->dev.Ft = h->d_Ft;
    h->dev.Ut = h->d_Ut;
    {   // multiply form for large batches (k_wide.hip): W = L^-1 as MFMA operand tiles
        std::vector<double> Wt, Wtb;
        mcd::pack_w_tiles(n, W, Wt, Wtb);
        HIP_TRY(hipMalloc((void**)&h->d_Wt, Wt.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(h->d_Wt, Wt.data(), Wt.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&h->d_Wtb, Wtb.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(h->d_Wtb, Wtb.data(), Wtb.size() * sizeof(double), hipMemcpyHostToDevice));
        h->dev.Wt = h->d_Wt;
        h->dev.Wtb = h->d_Wtb;
        if (h->R >= 1) {                     // columns of W for the incremental evaluation of sparse Metropolis-Hastings proposals
                                             // (k_mh_chain_big.hip at R <= 4, k_mh_inc.hip above)
            std::vector<double> Wc((size_t)n * NP, 0.0);
            for (int j = 0; j < n; ++j)
                for (int i = j; i < n; ++i) Wc[(size_t)j * NP + i] = W[(size_t)i * n + j];
            HIP_TRY(hipMalloc((void**)&h->d_Wc, Wc.size() * sizeof(double)));
            HIP_TRY(hipMemcpy(h->d_Wc, Wc.data(), Wc.size() * sizeof(double), hipMemcpyHostToDevice));
            h->dev.Wc = h->d_Wc;
        }
        HIP_TRY(mcd::prepare_wide());
        HIP_TRY(mcd::prepare_wide_grad());
        HIP_TRY(mcd::prepare_wide_grad_mc());
        if (n > 128) {                       // row-split form (k_split.hip): below, the sweep's short dependent chain wins
            hipError_t e = hipSuccess;
            h->split = mcd::split_host_create(n, W.data(), &e);
            HIP_TRY(e);
        }
        h->dev.split = h->split;
        h->dev.form = &h->form;
    }
    *out = h.release();
    return MCD_OK;
}

void mcd_mvn_destroy(mcd_mvn_t* h) { delete h; }

int mcd_set_logpdf_form(int form)
{
    if (form < MCD_FORM_AUTO || form > MCD_FORM_MULTIPLY) return mcd_set_last_error_(MCD_ERR_INVALID_ARG, "mcd_set_logpdf_form: unknown form");
    return mcd::set_logpdf_form(form);
}

int mcd_mvn_set_form(const mcd_mvn_t* h, int form)
{
    if (!h) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_set_form: NULL handle");
    if (form < MCD_FORM_AUTO || form > MCD_FORM_MULTIPLY) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_set_form: unknown form");
    return __atomic_exchange_n(&h->form, form, __ATOMIC_RELAXED);
}

int mcd_mvn_release_stream(const mcd_mvn_t* h, void* stream)
{
    if (!h) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_release_stream: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(mcd::split_release_stream(h->split, (hipStream_t)stream));
    return MCD_OK;
}

int mcd_mvn_dim(const mcd_mvn_t* h) { return h ? h->n : fail(MCD_ERR_INVALID_ARG, "mcd_mvn_dim: NULL handle"); }
int mcd_mvn_device(const mcd_mvn_t* h) { return h ? h->device : fail(MCD_ERR_INVALID_ARG, "mcd_mvn_device: NULL handle"); }
double mcd_mvn_logdet(const mcd_mvn_t* h) { return h ? h->logdet : std::nan(""); }

int mcd_mvn_get_factor(const mcd_mvn_t* h, double* L_out)
{
    if (!h || !L_out) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_get_factor: NULL argument");
    std::memcpy(L_out, h->L.data(), sizeof(double) * (size_t)h->n * h->n);
    return MCD_OK;
}

static int launch_logpdf_any(const mcd_mvn* h, const double* X, int64_t ld, int64_t batch, double* ll, hipStream_t st)
{
    HIP_TRY(mcd::launch_logpdf(h->dev, X, ld, batch, ll, st));
    return MCD_OK;
}

int mcd_mvn_logpdf_batch(const mcd_mvn_t* h, const double* X, int64_t ld, int64_t batch, int on_device, void* stream,
                         double* ll)
{
    if (!h) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_logpdf_batch: NULL handle");
    if (batch < 0 || ld < h->n) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_logpdf_batch: need batch >= 0 and ld >= n");
    if (batch == 0) return MCD_OK;
    if (!X || !ll) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_logpdf_batch: NULL data pointer");
    HIP_TRY(hipSetDevice(h->device));
    if (on_device) {
        if (int rc = launch_logpdf_any(h, X, ld, batch, ll, (hipStream_t)stream)) return rc;
        return MCD_OK;
    }
    WsGuard g{&h->pool, h->pool.acquire()};
    if (!g.w) return fail(MCD_ERR_HIP, "mcd_mvn_logpdf_batch: cannot create a HIP stream");
    const size_t nx = (size_t)batch * (size_t)h->n;
    if (nx + (size_t)batch <= kZeroCopyDoubles) {
        if (int rc = ensure_mapped(g.w, nx + (size_t)batch)) return rc;
        double* hX = g.w->hbuf;
        for (int64_t b = 0; b < batch; ++b) memcpy(hX + (size_t)b * h->n, X + (size_t)b * ld, sizeof(double) * (size_t)h->n);
        if (int rc = launch_logpdf_any(h, g.w->hbuf_dev, h->n, batch, g.w->hbuf_dev + nx, g.w->stream)) return rc;
        HIP_TRY(hipStreamSynchronize(g.w->stream));
        memcpy(ll, hX + nx, sizeof(double) * (size_t)batch);
        return MCD_OK;
    }
    if (int rc = ensure(g.w, nx + (size_t)batch)) return rc;
    double* dX = g.w->dbuf;
    double* dll = dX + nx;
    HIP_TRY(hipMemcpy2DAsync(dX, sizeof(double) * h->n, X, sizeof(double) * ld, sizeof(double) * h->n, (size_t)batch,
                             hipMemcpyHostToDevice, g.w->stream));
    if (int rc = launch_logpdf_any(h, dX, h->n, batch, dll, g.w->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(ll, dll, sizeof(double) * (size_t)batch, hipMemcpyDeviceToHost, g.w->stream));
    HIP_TRY(hipStreamSynchronize(g.w->stream));
    return MCD_OK;
}

int mcd_mvn_logpdf(const mcd_mvn_t* h, const double* x, double* ll)
{
    if (!h) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_logpdf: NULL handle");
    return mcd_mvn_logpdf_batch(h, x, h->n, 1, 0, nullptr, ll);
}

int mcd_mvn_grad_batch(const mcd_mvn_t* h, const double* X, int64_t ld, int64_t batch, int on_device, void* stream,
                       double* ll, double* G, int64_t ldg)
{
    if (!h) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_grad_batch: NULL handle");
    if (batch < 0 || ld < h->n || ldg < h->n)
        return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_grad_batch: need batch >= 0, ld >= n and ldg >= n");
    if (batch == 0) return MCD_OK;
    if (!X || !ll || !G) return fail(MCD_ERR_INVALID_ARG, "mcd_mvn_grad_batch: NULL data pointer");
    HIP_TRY(hipSetDevice(h->device));
    if (on_device) {
        HIP_TRY(mcd::launch_grad(h->dev, X, ld, batch, ll, G, ldg, (hipStream_t)stream));
        return MCD_OK;
    }
    WsGuard g{&h->pool, h->pool.acquire()};
    if (!g.w) return fail(MCD_ERR_HIP, "mcd_mvn_grad_batch: cannot create a HIP stream");
    const size_t nx = (size_t)batch * (size_t)h->n;
    if (int rc = ensure(g.w, 2 * nx + (size_t)batch)) return rc;
    double* dX = g.w->dbuf;
    double* dG = dX + nx;
    double* dll = dG + nx;
    HIP_TRY(hipMemcpy2DAsync(dX, sizeof(double) * h->n, X, sizeof(double) * ld, sizeof(double) * h->n, (size_t)batch,
                             hipMemcpyHostToDevice, g.w->stream));
    HIP_TRY(mcd::launch_grad(h->dev, dX, h->n, batch, dll, dG, h->n, g.w->stream));
    HIP_TRY(hipMemcpyAsync(ll, dll, sizeof(double) * (size_t)batch, hipMemcpyDeviceToHost, g.w->stream));
    HIP_TRY(hipMemcpy2DAsync(G, sizeof(double) * ldg, dG, sizeof(double) * h->n, sizeof(double) * h->n, (size_t)batch,
                             hipMemcpyDeviceToHost, g.w->stream));
    HIP_TRY(hipStreamSynchronize(g.w->stream));
    return MCD_OK;
}

int mcd_tree_create(mcd_tree_t** out, const mcd_mvn_t* h, int n_nodes, const int32_t* parent)
{
    if (!out) return fail(MCD_ERR_INVALID_ARG, "mcd_tree_create: out is NULL");
    *out = nullptr;
    if (!h || !parent) return fail(MCD_ERR_INVALID_ARG, "mcd_tree_create: NULL argument");
    if (n_nodes < 3 || parent[0] != -1)
        return fail(MCD_ERR_INVALID_ARG, "mcd_tree_create: need n_nodes >= 3 and parent[0] == -1");
    for (int v = 1; v < n_nodes; ++v)
        if (parent[v] < 0 || parent[v] >= v)
            return fail(MCD_ERR_INVALID_ARG, "mcd_tree_create: parent[%d] = %d violates pre-order numbering", v, parent[v]);
    // pre-order check: the parent of v must be the last still-open ancestor chain member
    {
        std::vector<int> stack{0};
        for (int v = 1; v < n_nodes; ++v) {
            while (!stack.empty() && stack.back() != parent[v]) stack.pop_back();
            if (stack.empty())
                return fail(MCD_ERR_INVALID_ARG, "mcd_tree_create: node %d is not numbered in pre-order", v);
            stack.push_back(v);
        }
    }
    std::vector<int> root_children;
    for (int v = 1; v < n_nodes; ++v)
        if (parent[v] == 0) root_children.push_back(v);
    if (root_children.size() != 2)  // app/Tools.hs:43
        return fail(MCD_ERR_ROOT_NOT_BIFURCATING, "getBranches: Root node is not bifurcating.");
    if (n_nodes - 2 != h->n)
        return fail(MCD_ERR_INVALID_ARG, "mcd_tree_create: tree has %d branches after merging the root branches, likelihood has dimension %d",
                    n_nodes - 2, h->n);
    const int l = root_children[0], r = root_children[1];  // l == 1 in pre-order
    // getBranches order: [l, r] ++ (l+1 .. r-1) ++ (r+1 .. n_nodes-1); sumFirstTwo merges the first two.
    const int NP = 64 * h->R;
    std::vector<int32_t> slot(NP, -1);
    slot[0] = l;
    int o = 1;
    for (int v = l + 1; v < r; ++v) slot[o++] = v;
    for (int v = r + 1; v < n_nodes; ++v) slot[o++] = v;
    slot.resize(2 * (size_t)NP, 0);                        // second half: the parent of each slot's node
    for (int i = 0; i < NP; ++i) slot[NP + i] = slot[i] >= 0 ? parent[slot[i]] : 0;
    std::vector<int32_t> cptr(n_nodes + 1, 0), cidx(n_nodes - 1);
    for (int v = 1; v < n_nodes; ++v) cptr[parent[v] + 1]++;
    for (int v = 0; v < n_nodes; ++v) cptr[v + 1] += cptr[v];
    {
        std::vector<int32_t> fill(cptr.begin(), cptr.end() - 1);
        for (int v = 1; v < n_nodes; ++v) cidx[fill[parent[v]]++] = v;  // ascending ids = left to right
    }
    std::unique_ptr<mcd_tree> t(new mcd_tree());
    t->mvn = h;
    t->n_nodes = n_nodes;
    t->parent.assign(parent, parent + n_nodes);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMalloc((void**)&t->d_parent, sizeof(int32_t) * n_nodes));
    HIP_TRY(hipMalloc((void**)&t->d_slot, sizeof(int32_t) * 2 * NP));
    HIP_TRY(hipMalloc((void**)&t->d_cptr, sizeof(int32_t) * (n_nodes + 1)));
    HIP_TRY(hipMalloc((void**)&t->d_cidx, sizeof(int32_t) * (n_nodes - 1)));
    HIP_TRY(hipMemcpy(t->d_parent, parent, sizeof(int32_t) * n_nodes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t->d_slot, slot.data(), sizeof(int32_t) * 2 * NP, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t->d_cptr, cptr.data(), sizeof(int32_t) * (n_nodes + 1), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t->d_cidx, cidx.data(), sizeof(int32_t) * (n_nodes - 1), hipMemcpyHostToDevice));
    t->dev.n_nodes = n_nodes;
    t->dev.n_nodes_pad = (n_nodes + 63) / 64 * 64;
    t->dev.root_right = r;
    t->dev.parent = t->d_parent;
    t->dev.slot_node = t->d_slot;
    t->dev.slot_parent = t->d_slot + NP;
    t->dev.child_ptr = t->d_cptr;
    t->dev.child_idx = t->d_cidx;
    *out = t.release();
    return MCD_OK;
}

void mcd_tree_destroy(mcd_tree_t* t) { delete t; }

int mcd_tree_n_nodes(const mcd_tree_t* t) { return t ? t->n_nodes : fail(MCD_ERR_INVALID_ARG, "mcd_tree_n_nodes: NULL handle"); }

int mcd_tree_loglik_batch(const mcd_tree_t* t, const double* heights, const double* rates, int64_t ld_state,
                          const double* tH, const double* rMu, int64_t batch, int on_device, void* stream, double* ll,
                          double* log_jac)
{
    if (!t) return fail(MCD_ERR_INVALID_ARG, "mcd_tree_loglik_batch: NULL handle");
    const mcd_mvn* h = t->mvn;
    if (batch < 0 || ld_state < t->n_nodes)
        return fail(MCD_ERR_INVALID_ARG, "mcd_tree_loglik_batch: need batch >= 0 and ld_state >= n_nodes");
    if (batch == 0) return MCD_OK;
    if (!heights || !rates || !tH || !rMu || !ll) return fail(MCD_ERR_INVALID_ARG, "mcd_tree_loglik_batch: NULL data pointer");
    HIP_TRY(hipSetDevice(h->device));
    if (on_device) {
        HIP_TRY(mcd::launch_tree_logpdf(h->dev, t->dev, heights, rates, ld_state, tH, rMu, batch, ll, log_jac,
                                        (hipStream_t)stream));
        return MCD_OK;
    }
    WsGuard g{&h->pool, h->pool.acquire()};
    if (!g.w) return fail(MCD_ERR_HIP, "mcd_tree_loglik_batch: cannot create a HIP stream");
    const size_t nn = (size_t)t->n_nodes, B = (size_t)batch;
    if (2 * B * nn + 4 * B <= kZeroCopyDoubles) {
        if (int rc = ensure_mapped(g.w, 2 * B * nn + 4 * B)) return rc;
        double* hH = g.w->hbuf;
        double* hR = hH + B * nn;
        double* hS = hR + B * nn;            // tH | rMu | ll | log_jac
        for (size_t b = 0; b < B; ++b) {
            memcpy(hH + b * nn, heights + b * (size_t)ld_state, sizeof(double) * nn);
            memcpy(hR + b * nn, rates + b * (size_t)ld_state, sizeof(double) * nn);
        }
        memcpy(hS, tH, sizeof(double) * B);
        memcpy(hS + B, rMu, sizeof(double) * B);
        double* d0 = g.w->hbuf_dev;
        HIP_TRY(mcd::launch_tree_logpdf(h->dev, t->dev, d0, d0 + B * nn, (int64_t)nn, d0 + 2 * B * nn, d0 + 2 * B * nn + B, batch,
                                        d0 + 2 * B * nn + 2 * B, log_jac ? d0 + 2 * B * nn + 3 * B : nullptr, g.w->stream));
        HIP_TRY(hipStreamSynchronize(g.w->stream));
        memcpy(ll, hS + 2 * B, sizeof(double) * B);
        if (log_jac) memcpy(log_jac, hS + 3 * B, sizeof(double) * B);
        return MCD_OK;
    }
    if (int rc = ensure(g.w, 2 * B * nn + 4 * B)) return rc;
    double* dH = g.w->dbuf;
    double* dR = dH + B * nn;
    double* dtH = dR + B * nn;
    double* drMu = dtH + B;
    double* dll = drMu + B;
    double* dlj = dll + B;
    hipStream_t st = g.w->stream;
    HIP_TRY(hipMemcpy2DAsync(dH, sizeof(double) * nn, heights, sizeof(double) * ld_state, sizeof(double) * nn, B,
                             hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(dR, sizeof(double) * nn, rates, sizeof(double) * ld_state, sizeof(double) * nn, B,
                             hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dtH, tH, sizeof(double) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(drMu, rMu, sizeof(double) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(mcd::launch_tree_logpdf(h->dev, t->dev, dH, dR, (int64_t)nn, dtH, drMu, batch, dll, log_jac ? dlj : nullptr, st));
    HIP_TRY(hipMemcpyAsync(ll, dll, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    if (log_jac) HIP_TRY(hipMemcpyAsync(log_jac, dlj, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MCD_OK;
}

int mcd_tree_grad_batch(const mcd_tree_t* t, const double* heights, const double* rates, int64_t ld_state,
                        const double* tH, const double* rMu, int64_t batch, int on_device, void* stream, double* ll,
                        double* g_heights, double* g_rates, double* g_tH, double* g_rMu)
{
    if (!t) return fail(MCD_ERR_INVALID_ARG, "mcd_tree_grad_batch: NULL handle");
    const mcd_mvn* h = t->mvn;
    if (batch < 0 || ld_state < t->n_nodes)
        return fail(MCD_ERR_INVALID_ARG, "mcd_tree_grad_batch: need batch >= 0 and ld_state >= n_nodes");
    if (batch == 0) return MCD_OK;
    if (!heights || !rates || !tH || !rMu || !ll || !g_heights || !g_rates || !g_tH || !g_rMu)
        return fail(MCD_ERR_INVALID_ARG, "mcd_tree_grad_batch: NULL data pointer");
    HIP_TRY(hipSetDevice(h->device));
    if (on_device) {
        HIP_TRY(mcd::launch_tree_grad(h->dev, t->dev, heights, rates, ld_state, tH, rMu, batch, ll, g_heights, g_rates,
                                      g_tH, g_rMu, (hipStream_t)stream));
        return MCD_OK;
    }
    WsGuard g{&h->pool, h->pool.acquire()};
    if (!g.w) return fail(MCD_ERR_HIP, "mcd_tree_grad_batch: cannot create a HIP stream");
    const size_t nn = (size_t)t->n_nodes, B = (size_t)batch;
    if (int rc = ensure(g.w, 4 * B * nn + 5 * B)) return rc;
    double* dH = g.w->dbuf;
    double* dR = dH + B * nn;
    double* dgH = dR + B * nn;
    double* dgR = dgH + B * nn;
    double* dtH = dgR + B * nn;
    double* drMu = dtH + B;
    double* dll = drMu + B;
    double* dgt = dll + B;
    double* dgm = dgt + B;
    hipStream_t st = g.w->stream;
    HIP_TRY(hipMemcpy2DAsync(dH, sizeof(double) * nn, heights, sizeof(double) * ld_state, sizeof(double) * nn, B,
                             hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(dR, sizeof(double) * nn, rates, sizeof(double) * ld_state, sizeof(double) * nn, B,
                             hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dtH, tH, sizeof(double) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(drMu, rMu, sizeof(double) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(mcd::launch_tree_grad(h->dev, t->dev, dH, dR, (int64_t)nn, dtH, drMu, batch, dll, dgH, dgR, dgt, dgm, st));
    HIP_TRY(hipMemcpyAsync(ll, dll, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpy2DAsync(g_heights, sizeof(double) * ld_state, dgH, sizeof(double) * nn, sizeof(double) * nn, B,
                             hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpy2DAsync(g_rates, sizeof(double) * ld_state, dgR, sizeof(double) * nn, sizeof(double) * nn, B,
                             hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(g_tH, dgt, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(g_rMu, dgm, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MCD_OK;
}

}  // extern "C"
