// sparse_capi.cpp -- the C ABI of the sparse form (include/mcmcdate_mvn.h: mcd_sparse_*) on top of k_sparse.hip.
// Replaces: likelihoodFunction (Sparse mu sigmaInvSparse logDetSigma) (app/Probability.hs:279, 178-184) with the operands
// of getData's SparseS branch (app/Main.hs:95-97).  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <memory>
#include <numeric>
#include <vector>

#include "../../include/mcmcdate_mvn.h"
#include "mvn_kernels.h"

extern "C" int mcd_set_last_error_(int code, const char* msg);

namespace {

int sfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return mcd_set_last_error_(code, buf);
}

#define SHIP_TRY(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return sfail(MCD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

}  // namespace

struct mcd_sparse {
    int n = 0, device = 0;
    mcd::SparseDev dev{};
    std::vector<void*> allocs;
    // scratch of the device-resident entry points (the transposed chain vectors, partial forms): one grow-only buffer per stream
    // of the caller's, launches on a stream being ordered.  (A stream under capture shares its eager buffer: a graph replayed on
    // ANOTHER stream must not run beside launches on the stream it was captured from.)
    mutable std::mutex mu;
    mutable std::map<hipStream_t, std::pair<double*, size_t>> scratch;
    ~mcd_sparse()
    {
        (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        for (auto& kv : scratch)
            if (kv.second.first) (void)hipFree(kv.second.first);
    }
};

struct mcd_sparse_tree {
    const mcd_sparse* sp = nullptr;
    mcd::SparseTreeDev dev{};
    std::vector<int32_t> parent;   // host copy (the Metropolis-Hastings driver checks its proposal table against the topology)
    int32_t* d_slot = nullptr;
    ~mcd_sparse_tree()
    {
        if (sp) (void)hipSetDevice(sp->device);
        if (d_slot) (void)hipFree(d_slot);
    }
};

namespace {

int scratch_for(const mcd_sparse* h, hipStream_t st, size_t doubles, double** out)
{
    std::lock_guard<std::mutex> lock(h->mu);
    auto& e = h->scratch[st];
    if (e.second < doubles) {
        // growing frees the old buffer: a graph captured on this stream earlier has its address baked in -- refuse while a capture is open
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone && e.first != nullptr)
            return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse: the stream's scratch would have to grow during a stream capture (make the largest call once before capturing)");
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;      // (legal while the calling thread captures a stream)
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e.first) (void)hipFree(e.first);                          // waits for the device: earlier launches are done with it
        e = {nullptr, 0};
        double* p = nullptr;
        const hipError_t err = hipMalloc((void**)&p, sizeof(double) * doubles);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (err != hipSuccess) return sfail(MCD_ERR_HIP, "mcd_sparse: %zu bytes of scratch: %s", sizeof(double) * doubles, hipGetErrorString(err));
        e = {p, doubles};
    }
    *out = e.first;
    return MCD_OK;
}

template <class T>
int upload(mcd_sparse* h, const T** dst, const T* src, size_t count)
{
    T* d = nullptr;
    SHIP_TRY(hipMalloc((void**)&d, sizeof(T) * (count ? count : 1)));
    h->allocs.push_back(d);
    if (count) SHIP_TRY(hipMemcpy(d, src, sizeof(T) * count, hipMemcpyHostToDevice));
    *dst = d;
    return MCD_OK;
}

// host arrays through a temporary device buffer on a stream of the call's own (host-pointer entry points: copy, run, copy)
struct Scratch {
    hipStream_t st = nullptr;
    std::vector<void*> bufs;
    ~Scratch()
    {
        for (void* p : bufs) (void)hipFree(p);
        if (st) (void)hipStreamDestroy(st);
    }
    double* alloc(size_t doubles)
    {
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(double) * (doubles ? doubles : 1)) != hipSuccess) return nullptr;
        bufs.push_back(p);
        return (double*)p;
    }
};

}  // namespace

// for the Metropolis-Hastings driver (mh_capi.cpp)
int mcd_sparse_tree_internal_(const mcd_sparse_tree* t, const mcd_sparse** sp, const mcd::SparseDev** dev, const mcd::SparseTreeDev** tree, int* device,
                              const int32_t** host_parent)
{
    if (!t || !t->sp) return MCD_ERR_INVALID_ARG;
    *sp = t->sp;
    *dev = &t->sp->dev;
    *tree = &t->dev;
    *device = t->sp->device;
    *host_parent = t->parent.data();
    return MCD_OK;
}
int mcd_sparse_scratch_(const mcd_sparse* h, hipStream_t st, int64_t batch, double** out)
{
    return scratch_for(h, st, mcd::sparse_scratch_doubles(h->n, batch, false), out);
}

extern "C" {

int mcd_sparse_create(mcd_sparse_t** out, int n, const double* mu, int64_t nnz, const int32_t* row, const int32_t* col, const double* val,
                      double logdet_sigma, int device_id)
{
    if (!out) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_create: out is NULL");
    *out = nullptr;
    if (n < 1 || n > mcd::kSparseMaxDim) return sfail(MCD_ERR_UNSUPPORTED, "mcd_sparse_create: dimension %d not in 1 .. %d", n, mcd::kSparseMaxDim);
    if (!mu || nnz < 0 || (nnz > 0 && (!row || !col || !val))) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_create: NULL argument");
    if (!std::isfinite(logdet_sigma)) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_create: logdet_sigma is not finite");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sfail(MCD_ERR_NO_DEVICE, "mcd_sparse_create: no HIP device (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_create: device %d of %d", device_id, ndev);
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(mu[i])) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_create: mu[%d] is not finite", i);
    for (int64_t k = 0; k < nnz; ++k) {
        if (row[k] < 0 || row[k] >= n || col[k] < 0 || col[k] >= n) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_create: entry %lld (%d, %d) outside the matrix", (long long)k, row[k], col[k]);
        if (!std::isfinite(val[k])) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_create: entry %lld is not finite", (long long)k);
    }
    // association list -> CSR (L.mkSparse, app/Main.hs:95): sorted by (row, column), entries of one position added up
    std::vector<int64_t> order((size_t)nnz);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return row[a] != row[b] ? row[a] < row[b] : col[a] < col[b]; });
    std::vector<int32_t> rowptr((size_t)n + 1, 0), cc, rr;
    std::vector<double> vv;
    for (size_t q = 0; q < order.size(); ++q) {
        const int64_t k = order[q];
        if (q > 0 && row[order[q - 1]] == row[k] && col[order[q - 1]] == col[k]) {
            vv.back() += val[k];
            continue;
        }
        cc.push_back(col[k]);
        rr.push_back(row[k]);
        vv.push_back(val[k]);
        rowptr[(size_t)row[k] + 1] += 1;
    }
    for (int i = 0; i < n; ++i) rowptr[(size_t)i + 1] += rowptr[(size_t)i];
    std::unique_ptr<mcd_sparse> h(new mcd_sparse());
    h->n = n;
    h->device = device_id;
    SHIP_TRY(hipSetDevice(device_id));
    int rc = MCD_OK;
    if ((rc = upload(h.get(), &h->dev.rowptr, rowptr.data(), rowptr.size())) || (rc = upload(h.get(), &h->dev.col, cc.data(), cc.size())) ||
        (rc = upload(h.get(), &h->dev.trow, rr.data(), rr.size())) ||
        (rc = upload(h.get(), &h->dev.val, vv.data(), vv.size())) || (rc = upload(h.get(), &h->dev.mu, mu, (size_t)n)))
        return rc;
    h->dev.n = n;
    h->dev.nnz = (int64_t)vv.size();
    {
        // Is the matrix exactly symmetric (pattern and values)?  `prepare`'s graphical-lasso estimate is (app/Main.hs:257-277); a hand-made
        // .data file need not be.  The symmetric part Ps = (P + P^T) / 2 in CSR serves the gradient -Ps dx and the Metropolis-Hastings
        // driver's incremental form (rows of the moved distances); for a symmetric matrix it IS the matrix (no second copy).
        const size_t nz = vv.size();
        bool symmetric = true;
        std::vector<int32_t> sym_rowptr, sym_col;                    // the symmetric part, when it is not the matrix itself
        std::vector<double> sym_val;
        auto find = [&](int r, int c) -> int64_t {
            const int32_t* lo = cc.data() + rowptr[(size_t)r];
            const int32_t* hi = cc.data() + rowptr[(size_t)r + 1];
            const int32_t* it = std::lower_bound(lo, hi, (int32_t)c);
            return (it != hi && *it == c) ? (int64_t)(it - cc.data()) : -1;
        };
        for (size_t e = 0; e < nz && symmetric; ++e) {
            const int64_t t = find(cc[e], rr[e]);
            symmetric = t >= 0 && vv[(size_t)t] == vv[e];
        }
        if (symmetric) {
            h->dev.s_nnz = (int64_t)nz;
            h->dev.s_rowptr = h->dev.rowptr;
            h->dev.s_trow = h->dev.trow;
            h->dev.s_col = h->dev.col;
            h->dev.s_val = h->dev.val;
        } else {
            // merge row i of P with column i of P (= row i of P^T): entries (i, k) -> (P[i][k] + P[k][i]) / 2
            std::vector<std::vector<std::pair<int32_t, double>>> rows((size_t)n);
            for (size_t e = 0; e < nz; ++e) {
                rows[(size_t)rr[e]].push_back({cc[e], 0.5 * vv[e]});
                rows[(size_t)cc[e]].push_back({rr[e], 0.5 * vv[e]});
            }
            std::vector<int32_t>& sp = sym_rowptr;
            std::vector<int32_t>& sc = sym_col;
            std::vector<int32_t> sr;
            std::vector<double>& sv = sym_val;
            sp.assign((size_t)n + 1, 0);
            for (int i = 0; i < n; ++i) {
                auto& r = rows[(size_t)i];
                std::stable_sort(r.begin(), r.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
                for (size_t q = 0; q < r.size(); ++q) {
                    if (q > 0 && r[q - 1].first == r[q].first) {
                        sv.back() += r[q].second;
                        continue;
                    }
                    sc.push_back(r[q].first);
                    sr.push_back(i);
                    sv.push_back(r[q].second);
                }
                sp[(size_t)i + 1] = (int32_t)sc.size();
            }
            h->dev.s_nnz = (int64_t)sv.size();
            if ((rc = upload(h.get(), &h->dev.s_rowptr, sp.data(), sp.size())) || (rc = upload(h.get(), &h->dev.s_col, sc.data(), sc.size())) ||
                (rc = upload(h.get(), &h->dev.s_trow, sr.data(), sr.size())) || (rc = upload(h.get(), &h->dev.s_val, sv.data(), sv.size())))
                return rc;
        }
        {   // the rows of the symmetric part as fixed-size records (SparseDev::ell_*)
            const std::vector<int32_t>& rp = symmetric ? rowptr : sym_rowptr;
            const std::vector<int32_t>& rcol = symmetric ? cc : sym_col;
            const std::vector<double>& rv = symmetric ? vv : sym_val;
            const int W = mcd::kSparseEllW;
            std::vector<int32_t> ec((size_t)n * W), em((size_t)n, 0);
            std::vector<double> ev((size_t)n * W, 0.0), eu((size_t)n * W, 0.0);
            for (int i = 0; i < n; ++i) {
                const int len = rp[(size_t)i + 1] - rp[(size_t)i];
                for (int u = 0; u < W; ++u) {
                    const bool in = u < len;
                    const int32_t k = in ? rcol[(size_t)rp[(size_t)i] + u] : (int32_t)i;      // (padding: the row's own index, weight 0)
                    ec[(size_t)i * W + u] = k;
                    ev[(size_t)i * W + u] = in ? rv[(size_t)rp[(size_t)i] + u] : 0.0;
                    eu[(size_t)i * W + u] = mu[k];
                }
                em[(size_t)i] = len > W ? len - W : 0;
            }
            if ((rc = upload(h.get(), &h->dev.ell_col, ec.data(), ec.size())) || (rc = upload(h.get(), &h->dev.ell_val, ev.data(), ev.size())) ||
                (rc = upload(h.get(), &h->dev.ell_mu, eu.data(), eu.size())) || (rc = upload(h.get(), &h->dev.ell_more, em.data(), em.size())))
                return rc;
        }
        // the flat entry stream of the one-launch form: (row | column << 16, value); a symmetric matrix as its upper triangle with the
        // off-diagonal values doubled
        if (n <= 65535) {
            std::vector<uint32_t> qrc;
            std::vector<double> qv;
            for (size_t e = 0; e < nz; ++e) {
                if (symmetric && cc[e] < rr[e]) continue;
                qrc.push_back((uint32_t)rr[e] | ((uint32_t)cc[e] << 16));
                qv.push_back((symmetric && cc[e] != rr[e]) ? 2.0 * vv[e] : vv[e]);
            }
            h->dev.q_nnz = (int64_t)qv.size();
            if ((rc = upload(h.get(), &h->dev.q_rc, qrc.data(), qrc.size())) || (rc = upload(h.get(), &h->dev.q_val, qv.data(), qv.size()))) return rc;
        }
    }
    h->dev.c = -(0.9189385332046727417803297364056176 * (double)n);   // m_ln_sqrt_2_pi * k, Probability.hs:181-183
    h->dev.logdet = logdet_sigma;
    *out = h.release();
    return MCD_OK;
}

void mcd_sparse_destroy(mcd_sparse_t* h) { delete h; }

int mcd_sparse_release_stream(const mcd_sparse_t* h, void* stream)
{
    if (!h) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_release_stream: NULL handle");
    SHIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lock(h->mu);
    auto it = h->scratch.find((hipStream_t)stream);
    if (it == h->scratch.end()) return MCD_OK;
    SHIP_TRY(hipStreamSynchronize((hipStream_t)stream));    // the launches that use the buffer have run
    if (it->second.first) (void)hipFree(it->second.first);
    h->scratch.erase(it);
    return MCD_OK;
}
int mcd_sparse_dim(const mcd_sparse_t* h) { return h ? h->n : sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_dim: NULL handle"); }
int64_t mcd_sparse_nnz(const mcd_sparse_t* h) { return h ? h->dev.nnz : (int64_t)sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_nnz: NULL handle"); }

int mcd_sparse_logpdf_batch(const mcd_sparse_t* h, const double* X, int64_t ld, int64_t batch, int on_device, void* stream, double* ll)
{
    if (!h) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_logpdf_batch: NULL handle");
    if (batch < 0 || ld < h->n) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_logpdf_batch: need batch >= 0 and ld >= n");
    if (batch == 0) return MCD_OK;
    if (!X || !ll) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_logpdf_batch: NULL data pointer");
    SHIP_TRY(hipSetDevice(h->device));
    if (on_device) {
        double* sc = nullptr;
        if (int rc = scratch_for(h, (hipStream_t)stream, mcd::sparse_scratch_doubles(h->n, batch, false), &sc)) return rc;
        SHIP_TRY(mcd::launch_sparse_logpdf(h->dev, X, ld, batch, ll, sc, (hipStream_t)stream));
        return MCD_OK;
    }
    Scratch s;
    SHIP_TRY(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    double* dX = s.alloc((size_t)batch * h->n);
    double* dll = s.alloc((size_t)batch);
    double* sc = s.alloc(mcd::sparse_scratch_doubles(h->n, batch, false));
    if (!dX || !dll || !sc) return sfail(MCD_ERR_HIP, "mcd_sparse_logpdf_batch: out of device memory");
    SHIP_TRY(hipMemcpy2DAsync(dX, sizeof(double) * h->n, X, sizeof(double) * ld, sizeof(double) * h->n, (size_t)batch, hipMemcpyHostToDevice, s.st));
    SHIP_TRY(mcd::launch_sparse_logpdf(h->dev, dX, h->n, batch, dll, sc, s.st));
    SHIP_TRY(hipMemcpyAsync(ll, dll, sizeof(double) * (size_t)batch, hipMemcpyDeviceToHost, s.st));
    SHIP_TRY(hipStreamSynchronize(s.st));
    return MCD_OK;
}

int mcd_sparse_grad_batch(const mcd_sparse_t* h, const double* X, int64_t ld, int64_t batch, int on_device, void* stream, double* ll, double* G,
                          int64_t ldg)
{
    if (!h) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_grad_batch: NULL handle");
    if (batch < 0 || ld < h->n || ldg < h->n) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_grad_batch: need batch >= 0, ld >= n and ldg >= n");
    if (batch == 0) return MCD_OK;
    if (!X || !ll || !G) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_grad_batch: NULL data pointer");
    SHIP_TRY(hipSetDevice(h->device));
    if (on_device) {
        double* sc = nullptr;
        if (int rc = scratch_for(h, (hipStream_t)stream, mcd::sparse_scratch_doubles(h->n, batch, true), &sc)) return rc;
        SHIP_TRY(mcd::launch_sparse_grad(h->dev, X, ld, batch, ll, G, ldg, sc, (hipStream_t)stream));   // (G may be X: the stage has read every x)
        return MCD_OK;
    }
    Scratch s;
    SHIP_TRY(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    double* dX = s.alloc((size_t)batch * h->n);
    double* dG = s.alloc((size_t)batch * h->n);
    double* dll = s.alloc((size_t)batch);
    double* sc = s.alloc(mcd::sparse_scratch_doubles(h->n, batch, true));
    if (!dX || !dG || !dll || !sc) return sfail(MCD_ERR_HIP, "mcd_sparse_grad_batch: out of device memory");
    SHIP_TRY(hipMemcpy2DAsync(dX, sizeof(double) * h->n, X, sizeof(double) * ld, sizeof(double) * h->n, (size_t)batch, hipMemcpyHostToDevice, s.st));
    SHIP_TRY(mcd::launch_sparse_grad(h->dev, dX, h->n, batch, dll, dG, h->n, sc, s.st));
    SHIP_TRY(hipMemcpyAsync(ll, dll, sizeof(double) * (size_t)batch, hipMemcpyDeviceToHost, s.st));
    SHIP_TRY(hipMemcpy2DAsync(G, sizeof(double) * ldg, dG, sizeof(double) * h->n, sizeof(double) * h->n, (size_t)batch, hipMemcpyDeviceToHost, s.st));
    SHIP_TRY(hipStreamSynchronize(s.st));
    return MCD_OK;
}

int mcd_sparse_tree_create(mcd_sparse_tree_t** out, const mcd_sparse_t* h, int n_nodes, const int32_t* parent)
{
    if (!out) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_create: out is NULL");
    *out = nullptr;
    if (!h || !parent || n_nodes < 3) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_create: NULL argument or fewer than 3 nodes");
    if (parent[0] != -1) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_create: parent[0] must be -1 (root first, pre-order)");
    for (int v = 1; v < n_nodes; ++v)
        if (parent[v] < 0 || parent[v] >= v) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_create: nodes must be numbered in pre-order (parent[%d] = %d)", v, parent[v]);
    std::vector<int> rc;
    for (int v = 1; v < n_nodes; ++v)
        if (parent[v] == 0) rc.push_back(v);
    if (rc.size() != 2) return sfail(MCD_ERR_ROOT_NOT_BIFURCATING, "getBranches: Root node is not bifurcating.");   // app/Tools.hs:43
    if (n_nodes - 2 != h->n)
        return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_create: tree has %d branches after merging the root branches, likelihood has dimension %d",
                     n_nodes - 2, h->n);
    const int l = rc[0], r = rc[1];
    // getBranches order: [l, r] ++ (l+1 .. r-1) ++ (r+1 .. n_nodes-1); sumFirstTwo merges the first two (app/Tools.hs:36-48)
    std::vector<int32_t> slot;
    slot.push_back(l);
    for (int v = l + 1; v < r; ++v) slot.push_back(v);
    for (int v = r + 1; v < n_nodes; ++v) slot.push_back(v);
    const size_t n = slot.size();
    slot.resize(2 * n);
    for (size_t i = 0; i < n; ++i) slot[n + i] = parent[slot[i]];
    std::unique_ptr<mcd_sparse_tree> t(new mcd_sparse_tree());
    t->sp = h;
    t->parent.assign(parent, parent + n_nodes);
    SHIP_TRY(hipSetDevice(h->device));
    SHIP_TRY(hipMalloc((void**)&t->d_slot, sizeof(int32_t) * slot.size()));
    SHIP_TRY(hipMemcpy(t->d_slot, slot.data(), sizeof(int32_t) * slot.size(), hipMemcpyHostToDevice));
    t->dev.n_nodes = n_nodes;
    t->dev.root_right = r;
    t->dev.slot_node = t->d_slot;
    t->dev.slot_parent = t->d_slot + n;
    *out = t.release();
    return MCD_OK;
}

void mcd_sparse_tree_destroy(mcd_sparse_tree_t* t) { delete t; }

int mcd_sparse_tree_loglik_batch(const mcd_sparse_tree_t* t, const double* heights, const double* rates, int64_t ld_state, const double* tH,
                                 const double* rMu, int64_t batch, int on_device, void* stream, double* ll, double* log_jac)
{
    if (!t) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_loglik_batch: NULL handle");
    const int nn = t->dev.n_nodes;
    if (batch < 0 || ld_state < nn) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_loglik_batch: need batch >= 0 and ld_state >= n_nodes");
    if (batch == 0) return MCD_OK;
    if (!heights || !rates || !tH || !rMu || !ll) return sfail(MCD_ERR_INVALID_ARG, "mcd_sparse_tree_loglik_batch: NULL data pointer");
    const mcd_sparse* h = t->sp;
    SHIP_TRY(hipSetDevice(h->device));
    if (on_device) {
        double* sc = nullptr;
        if (int rc = scratch_for(h, (hipStream_t)stream, mcd::sparse_scratch_doubles(h->n, batch, false), &sc)) return rc;
        SHIP_TRY(mcd::launch_sparse_tree_logpdf(h->dev, t->dev, heights, rates, ld_state, tH, rMu, batch, ll, log_jac, sc, (hipStream_t)stream));
        return MCD_OK;
    }
    Scratch s;
    SHIP_TRY(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    const size_t BN = (size_t)batch * nn, B = (size_t)batch;
    double* dH = s.alloc(BN);
    double* dR = s.alloc(BN);
    double* dsc = s.alloc(4 * B);
    double* sc = s.alloc(mcd::sparse_scratch_doubles(h->n, batch, false));
    if (!dH || !dR || !dsc || !sc) return sfail(MCD_ERR_HIP, "mcd_sparse_tree_loglik_batch: out of device memory");
    SHIP_TRY(hipMemcpy2DAsync(dH, sizeof(double) * nn, heights, sizeof(double) * ld_state, sizeof(double) * nn, B, hipMemcpyHostToDevice, s.st));
    SHIP_TRY(hipMemcpy2DAsync(dR, sizeof(double) * nn, rates, sizeof(double) * ld_state, sizeof(double) * nn, B, hipMemcpyHostToDevice, s.st));
    SHIP_TRY(hipMemcpyAsync(dsc, tH, sizeof(double) * B, hipMemcpyHostToDevice, s.st));
    SHIP_TRY(hipMemcpyAsync(dsc + B, rMu, sizeof(double) * B, hipMemcpyHostToDevice, s.st));
    SHIP_TRY(mcd::launch_sparse_tree_logpdf(h->dev, t->dev, dH, dR, nn, dsc, dsc + B, batch, dsc + 2 * B, dsc + 3 * B, sc, s.st));
    SHIP_TRY(hipMemcpyAsync(ll, dsc + 2 * B, sizeof(double) * B, hipMemcpyDeviceToHost, s.st));
    if (log_jac) SHIP_TRY(hipMemcpyAsync(log_jac, dsc + 3 * B, sizeof(double) * B, hipMemcpyDeviceToHost, s.st));
    SHIP_TRY(hipStreamSynchronize(s.st));
    return MCD_OK;
}

}  // extern "C"
