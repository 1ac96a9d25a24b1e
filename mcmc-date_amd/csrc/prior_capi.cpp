// prior_capi.cpp -- C ABI of the batched prior (include/mcmcdate_mvn.h, "Prior" section).  No CPU path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mcmcdate_mvn.h"
#include "mvn_kernels.h"

extern "C" int mcd_set_last_error_(int code, const char* msg);   // mvn_capi.cpp

namespace {

int pfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return mcd_set_last_error_(code, buf);
}

#define PHIP_TRY(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return pfail(MCD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

template <class T>
int upload(const std::vector<T>& v, T** d)
{
    *d = nullptr;
    const size_t bytes = sizeof(T) * (v.empty() ? 1 : v.size());
    PHIP_TRY(hipMalloc((void**)d, bytes));
    if (!v.empty()) PHIP_TRY(hipMemcpy(*d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return MCD_OK;
}

struct Scratch {
    hipStream_t stream = nullptr;
    double* buf = nullptr;
    size_t cap = 0;
};

}  // namespace

struct mcd_prior {
    int device = 0, n_nodes = 0;
    mcd::PriorDev dev{};
    std::vector<void*> allocs;
    std::mutex mu;
    std::vector<Scratch*> idle;

    ~mcd_prior()
    {
        (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        for (Scratch* s : idle) {
            if (s->buf) (void)hipFree(s->buf);
            if (s->stream) (void)hipStreamDestroy(s->stream);
            delete s;
        }
    }
};

int mcd_prior_internal_(const mcd_prior* p, const mcd::PriorDev** prior, int* device)
{
    if (!p) return MCD_ERR_INVALID_ARG;
    *prior = &p->dev;
    *device = p->device;
    return MCD_OK;
}

extern "C" {

int mcd_prior_create(mcd_prior_t** out, int n_nodes, const int32_t* parent, double ht, int clock_model, int n_cal,
                     const int32_t* cal_node, const int32_t* cal_has_lo, const double* cal_lo, const double* cal_lo_p,
                     const int32_t* cal_has_hi, const double* cal_hi, const double* cal_hi_p, int n_con,
                     const int32_t* con_young, const int32_t* con_old, const double* con_p, int n_brace,
                     const int32_t* brace_ptr, const int32_t* brace_nodes, const double* brace_sd, int device_id)
{
    if (!out) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: out is NULL");
    *out = nullptr;
    if (n_nodes < 3 || !parent || parent[0] != -1) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: need n_nodes >= 3 and parent[0] == -1");
    if (!(ht > 0) || !std::isfinite(ht)) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: ht must be positive");   // exponential: rate <= 0 is an `error`
    if (clock_model < 0 || clock_model > 3) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: unknown relaxed molecular clock model %d", clock_model);
    if (n_cal < 0 || n_con < 0 || n_brace < 0) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: negative count");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return pfail(MCD_ERR_NO_DEVICE, "mcd_prior_create: no HIP device available (this library has no CPU path)");
    if (device_id < 0 || device_id >= ndev) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: device_id %d out of range", device_id);
    std::vector<int32_t> par(parent, parent + n_nodes), first(n_nodes, -1), second(n_nodes, -1), nch(n_nodes, 0);
    {
        std::vector<int> stack{0};
        for (int v = 1; v < n_nodes; ++v) {
            if (par[v] < 0 || par[v] >= v) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: parent[%d] violates pre-order numbering", v);
            while (!stack.empty() && stack.back() != par[v]) stack.pop_back();
            if (stack.empty()) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: node %d is not numbered in pre-order", v);
            stack.push_back(v);
            if (nch[par[v]] == 0) first[par[v]] = v;
            if (nch[par[v]] == 1) second[par[v]] = v;
            nch[par[v]]++;
        }
    }
    if (nch[0] != 2) return pfail(MCD_ERR_ROOT_NOT_BIFURCATING, "birthDeath: Tree is not bifurcating.");   // BirthDeath.hs:177
    for (int v = 1; v < n_nodes; ++v)
        if (nch[v] > 2) return pfail(MCD_ERR_INVALID_ARG, "birthDeathWith: Tree is multifurcating.");       // :232
    auto node_ok = [&](int v) { return v >= 0 && v < n_nodes; };
    for (int i = 0; i < n_cal; ++i) {
        if (!node_ok(cal_node[i])) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: calibration %d: bad node index", i);
        if (cal_has_lo[i] && !(cal_lo_p[i] > 0 && cal_lo_p[i] < 1)) return pfail(MCD_ERR_INVALID_ARG, "probabilityMass: calibration %d lower", i);
        if (cal_has_hi[i] && !(cal_hi_p[i] > 0 && cal_hi_p[i] < 1)) return pfail(MCD_ERR_INVALID_ARG, "probabilityMass: calibration %d upper", i);
    }
    for (int i = 0; i < n_con; ++i)
        if (!node_ok(con_young[i]) || !node_ok(con_old[i]) || !(con_p[i] > 0 && con_p[i] < 1))
            return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: constraint %d is invalid", i);
    for (int i = 0; i < n_brace; ++i) {
        if (!(brace_sd[i] > 0)) return pfail(MCD_ERR_INVALID_ARG, "braceSoftF: Standard deviation is zero or negative.");   // Brace.hs:223
        if (brace_ptr[i + 1] - brace_ptr[i] < 2) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: brace %d has fewer than two nodes", i);
        for (int j = brace_ptr[i]; j < brace_ptr[i + 1]; ++j)
            if (!node_ok(brace_nodes[j])) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_create: brace %d: bad node index", i);
    }
    std::unique_ptr<mcd_prior> p(new mcd_prior());
    p->device = device_id;
    p->n_nodes = n_nodes;
    PHIP_TRY(hipSetDevice(device_id));
    auto up_i = [&](const int32_t* src, int n, const int32_t** dst) -> int {
        int32_t* d;
        if (int rc = upload(std::vector<int32_t>(src, src + (n > 0 ? n : 0)), &d)) return rc;
        p->allocs.push_back(d);
        *dst = d;
        return MCD_OK;
    };
    auto up_d = [&](const double* src, int n, const double** dst) -> int {
        double* d;
        if (int rc = upload(std::vector<double>(src, src + (n > 0 ? n : 0)), &d)) return rc;
        p->allocs.push_back(d);
        *dst = d;
        return MCD_OK;
    };
    mcd::PriorDev& D = p->dev;
    D.n_nodes = n_nodes;
    D.clock_model = clock_model;
    D.ht = ht;
    D.n_cal = n_cal;
    D.n_con = n_con;
    D.n_brace = n_brace;
    int rc = MCD_OK;
    if ((rc = up_i(par.data(), n_nodes, &D.parent)) || (rc = up_i(first.data(), n_nodes, &D.first_child)) ||
        (rc = up_i(nch.data(), n_nodes, &D.n_children)) || (rc = up_i(second.data(), n_nodes, &D.second_child)) || (rc = up_i(cal_node, n_cal, &D.cal_node)) ||
        (rc = up_i(cal_has_lo, n_cal, &D.cal_has_lo)) || (rc = up_i(cal_has_hi, n_cal, &D.cal_has_hi)) ||
        (rc = up_d(cal_lo, n_cal, &D.cal_lo)) || (rc = up_d(cal_lo_p, n_cal, &D.cal_lo_p)) ||
        (rc = up_d(cal_hi, n_cal, &D.cal_hi)) || (rc = up_d(cal_hi_p, n_cal, &D.cal_hi_p)) ||
        (rc = up_i(con_young, n_con, &D.con_young)) || (rc = up_i(con_old, n_con, &D.con_old)) ||
        (rc = up_d(con_p, n_con, &D.con_p)) || (rc = up_i(brace_ptr, n_brace > 0 ? n_brace + 1 : 0, &D.brace_ptr)) ||
        (rc = up_i(brace_nodes, n_brace > 0 ? brace_ptr[n_brace] : 0, &D.brace_nodes)) ||
        (rc = up_d(brace_sd, n_brace, &D.brace_sd)))
        return rc;
    *out = p.release();
    return MCD_OK;
}

void mcd_prior_destroy(mcd_prior_t* p) { delete p; }

int mcd_prior_logprior_batch(const mcd_prior_t* cp, const double* birth, const double* death, const double* tH,
                             const double* heights, const double* rMu, const double* rVar, const double* rates,
                             int64_t ld_state, int64_t batch, int on_device, void* stream, double* lp, double* components)
{
    if (!cp) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_logprior_batch: NULL handle");
    mcd_prior* p = const_cast<mcd_prior*>(cp);
    if (batch < 0 || ld_state < p->n_nodes) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_logprior_batch: need batch >= 0 and ld_state >= n_nodes");
    if (batch == 0) return MCD_OK;
    if (!birth || !death || !tH || !heights || !rMu || !rVar || !rates || !lp)
        return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_logprior_batch: NULL data pointer");
    PHIP_TRY(hipSetDevice(p->device));
    if (on_device) {
        PHIP_TRY(mcd::launch_prior(p->dev, birth, death, tH, heights, rMu, rVar, rates, ld_state, batch, lp, components,
                                   (hipStream_t)stream));
        return MCD_OK;
    }
    Scratch* s = nullptr;
    {
        std::lock_guard<std::mutex> g(p->mu);
        if (!p->idle.empty()) {
            s = p->idle.back();
            p->idle.pop_back();
        }
    }
    if (!s) {
        s = new Scratch();
        if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
            delete s;
            return pfail(MCD_ERR_HIP, "mcd_prior_logprior_batch: cannot create a HIP stream");
        }
    }
    struct Back {
        mcd_prior* p;
        Scratch* s;
        ~Back()
        {
            std::lock_guard<std::mutex> g(p->mu);
            p->idle.push_back(s);
        }
    } back{p, s};
    const size_t B = (size_t)batch, nn = (size_t)p->n_nodes;
    const size_t need = 2 * B * nn + 5 * B + 4 * B;
    if (s->cap < need) {
        if (s->buf) PHIP_TRY(hipFree(s->buf));
        s->buf = nullptr;
        s->cap = 0;
        PHIP_TRY(hipMalloc((void**)&s->buf, (need + need / 4) * sizeof(double)));
        s->cap = need + need / 4;
    }
    double* dH = s->buf;
    double* dR = dH + B * nn;
    double* dsc = dR + B * nn;          // birth, death, tH, rMu, rVar
    double* dlp = dsc + 5 * B;
    double* dcomp = dlp + B;
    hipStream_t st = s->stream;
    PHIP_TRY(hipMemcpy2DAsync(dH, sizeof(double) * nn, heights, sizeof(double) * ld_state, sizeof(double) * nn, B, hipMemcpyHostToDevice, st));
    PHIP_TRY(hipMemcpy2DAsync(dR, sizeof(double) * nn, rates, sizeof(double) * ld_state, sizeof(double) * nn, B, hipMemcpyHostToDevice, st));
    const double* src[5] = {birth, death, tH, rMu, rVar};
    for (int i = 0; i < 5; ++i) PHIP_TRY(hipMemcpyAsync(dsc + i * B, src[i], sizeof(double) * B, hipMemcpyHostToDevice, st));
    PHIP_TRY(mcd::launch_prior(p->dev, dsc, dsc + B, dsc + 2 * B, dH, dsc + 3 * B, dsc + 4 * B, dR, (int64_t)nn, batch, dlp,
                               components ? dcomp : nullptr, st));
    PHIP_TRY(hipMemcpyAsync(lp, dlp, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    if (components) PHIP_TRY(hipMemcpyAsync(components, dcomp, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
    PHIP_TRY(hipStreamSynchronize(st));
    return MCD_OK;
}

int mcd_prior_grad_batch(const mcd_prior_t* cp, const double* birth, const double* death, const double* tH, const double* heights,
                         const double* rMu, const double* rVar, const double* rates, int64_t ld_state, int64_t batch, int on_device,
                         void* stream, double* lp, double* g_birth, double* g_death, double* g_tH, double* g_heights, double* g_rMu,
                         double* g_rVar, double* g_rates)
{
    if (!cp) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_grad_batch: NULL handle");
    mcd_prior* p = const_cast<mcd_prior*>(cp);
    if (batch < 0 || ld_state < p->n_nodes) return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_grad_batch: need batch >= 0 and ld_state >= n_nodes");
    if (batch == 0) return MCD_OK;
    if (!birth || !death || !tH || !heights || !rMu || !rVar || !rates || !lp || !g_birth || !g_death || !g_tH || !g_heights || !g_rMu ||
        !g_rVar || !g_rates)
        return pfail(MCD_ERR_INVALID_ARG, "mcd_prior_grad_batch: NULL data pointer");
    if ((size_t)p->n_nodes * 16 > 64 * 1024) return pfail(MCD_ERR_UNSUPPORTED, "mcd_prior_grad_batch: more than 4096 nodes");
    PHIP_TRY(hipSetDevice(p->device));
    if (on_device) {
        PHIP_TRY(mcd::launch_prior_grad(p->dev, birth, death, tH, heights, rMu, rVar, rates, ld_state, batch, lp, g_birth, g_death, g_tH,
                                        g_heights, g_rMu, g_rVar, g_rates, (hipStream_t)stream));
        return MCD_OK;
    }
    // host pointers: one private allocation per call (this entry point is not on a hot path of its own)
    const size_t B = (size_t)batch, nn = (size_t)p->n_nodes;
    const size_t need = 4 * B * nn + 11 * B;
    double* buf = nullptr;
    hipStream_t st = nullptr;
    PHIP_TRY(hipMalloc((void**)&buf, need * sizeof(double)));
    struct Free {
        double* b;
        hipStream_t* s;
        ~Free()
        {
            (void)hipFree(b);
            if (*s) (void)hipStreamDestroy(*s);
        }
    } fr{buf, &st};
    PHIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double* dH = buf;
    double* dR = dH + B * nn;
    double* dgH = dR + B * nn;
    double* dgR = dgH + B * nn;
    double* dsc = dgR + B * nn;          // birth, death, tH, rMu, rVar | lp | g_birth, g_death, g_tH, g_rMu, g_rVar
    PHIP_TRY(hipMemcpy2DAsync(dH, sizeof(double) * nn, heights, sizeof(double) * ld_state, sizeof(double) * nn, B, hipMemcpyHostToDevice, st));
    PHIP_TRY(hipMemcpy2DAsync(dR, sizeof(double) * nn, rates, sizeof(double) * ld_state, sizeof(double) * nn, B, hipMemcpyHostToDevice, st));
    const double* src[5] = {birth, death, tH, rMu, rVar};
    for (int i = 0; i < 5; ++i) PHIP_TRY(hipMemcpyAsync(dsc + i * B, src[i], sizeof(double) * B, hipMemcpyHostToDevice, st));
    double* o = dsc + 5 * B;
    PHIP_TRY(mcd::launch_prior_grad(p->dev, dsc, dsc + B, dsc + 2 * B, dH, dsc + 3 * B, dsc + 4 * B, dR, (int64_t)nn, batch, o, o + B, o + 2 * B,
                                    o + 3 * B, dgH, o + 4 * B, o + 5 * B, dgR, st));
    double* dst[6] = {lp, g_birth, g_death, g_tH, g_rMu, g_rVar};
    for (int i = 0; i < 6; ++i) PHIP_TRY(hipMemcpyAsync(dst[i], o + i * B, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    PHIP_TRY(hipMemcpy2DAsync(g_heights, sizeof(double) * ld_state, dgH, sizeof(double) * nn, sizeof(double) * nn, B, hipMemcpyDeviceToHost, st));
    PHIP_TRY(hipMemcpy2DAsync(g_rates, sizeof(double) * ld_state, dgR, sizeof(double) * nn, sizeof(double) * nn, B, hipMemcpyDeviceToHost, st));
    PHIP_TRY(hipStreamSynchronize(st));
    return MCD_OK;
}

}  // extern "C"
