// mh_capi.cpp -- C ABI of the lock-step Metropolis-Hastings-Green driver (include/mcmcdate_mvn.h, "mcd_mh_*").
// Trees of at most 64 nodes: the whole schedule in one launch (k_mh_chain.hip).  Larger trees, two launches per step:
// [accept the previous step + propose + ln prior] (k_mh.hip) and [batched likelihood + root-branch Jacobian]
// (k_tree_logpdf.hip), enqueued on one stream; the state stays on the device.  No CPU path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <algorithm>
#include <vector>

#include "../../include/mcmcdate_mvn.h"
#include "mvn_kernels.h"
#include "options.h"

extern "C" int mcd_set_last_error_(int code, const char* msg);   // mvn_capi.cpp
struct mcd_sparse;
struct mcd_sparse_tree;
int mcd_sparse_tree_internal_(const mcd_sparse_tree* t, const mcd_sparse** sp, const mcd::SparseDev** dev, const mcd::SparseTreeDev** tree, int* device,
                              const int32_t** host_parent);   // sparse_capi.cpp
int mcd_sparse_scratch_(const mcd_sparse* h, hipStream_t st, int64_t batch, double** out);

namespace {

int mfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return mcd_set_last_error_(code, buf);
}

#define MHIP_TRY(expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return mfail(MCD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

}  // namespace

struct mcd_mh {
    int device = 0;
    const mcd::MvnDev* mvn = nullptr;
    const mcd::TreeDev* tree = nullptr;
    const mcd::PriorDev* prior = nullptr;
    // a likelihood over a SPARSE precision matrix instead (mcd_mh_create_sparse): mvn stays null, tree points at tree_shim (the slot
    // tables the step kernel needs for the distances)
    const mcd_sparse* sp_handle = nullptr;
    const mcd::SparseDev* sp = nullptr;
    const mcd::SparseTreeDev* sp_tree = nullptr;
    mcd::TreeDev tree_shim{};
    mcd::MhDev dev{};
    uint64_t seed = 0, step = 0;
    int64_t n_samples = 0;
    bool have_state = false;
    bool chain_kernel = false;   // n_nodes <= 64: whole schedule in one launch
    double* d_X1 = nullptr;         // [batch][n]: distances of the proposed states (large trees: written by k_mh_step_wg)
    double* d_inc_ll = nullptr;     // [batch] ln likelihood output of the refreshing full products (not used)
    mcd::MhInc inc{};               // incremental likelihood of that path (k_mh_inc.hip): X0, zcur, zprop allocated on first use
    std::vector<mcd::MhRow> rows;   // host copy of the proposal table
    double* d_psum = nullptr;           // k_mh_step_wg's kept summands of the ln prior (MhDev::psum, psel)
    int32_t* d_psel = nullptr;
    std::vector<int32_t> sparse_rows;   // per row: 1 = moves at most kMhIncSlots distances (the two-launch path's incremental evaluation)
    const double* d_Fp = nullptr;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    int32_t* d_sched = nullptr;
    size_t sched_cap = 0;
    double* d_trace_alpha = nullptr;
    int8_t* d_trace_accept = nullptr;
    size_t trace_cap = 0;
    int last_path = MCD_MH_PATH_NONE;   // which launch structure the last mcd_mh_run took
    unsigned long long last_lds = 0;    // LDS bytes per workgroup of the persistent kernel that run launched last (0: none)
    bool list_all = false;              // sparse likelihood on a tree whose distance slots all fit the segment kernel's list
    // Metropolis-coupled MCMC (mcd_mh_mc3_*): temperature ranks of all GLOBAL chains, ladder, counters; phase = swap phases done
    mcd::Mc3Dev mc3{};
    uint64_t mc3_seed = 0, mc3_phase = 0;

    ~mcd_mh()
    {
        (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        if (d_sched) (void)hipFree(d_sched);
        if (d_trace_alpha) (void)hipFree(d_trace_alpha);
        if (d_trace_accept) (void)hipFree(d_trace_accept);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

template <class T>
int dev_alloc(mcd_mh* m, T** p, size_t count, bool zero)
{
    *p = nullptr;
    MHIP_TRY(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
    m->allocs.push_back(*p);
    if (zero) MHIP_TRY(hipMemset(*p, 0, sizeof(T) * (count ? count : 1)));
    return MCD_OK;
}

template <class T>
int dev_upload(mcd_mh* m, const T** p, const T* src, size_t count)
{
    T* d = nullptr;
    if (int rc = dev_alloc(m, &d, count, false)) return rc;
    if (count) MHIP_TRY(hipMemcpy(d, src, sizeof(T) * count, hipMemcpyHostToDevice));
    *p = d;
    return MCD_OK;
}

// ln prior, ln likelihood and ln jacobianRootBranch of a state batch -> post[3][batch]
int eval_posterior(mcd_mh* m, const double* sc, const double* H, const double* R, double* post)
{
    const mcd::MhDev& D = m->dev;
    const int64_t B = D.batch;
    MHIP_TRY(mcd::launch_prior(*m->prior, sc + 0 * B, sc + 1 * B, sc + 2 * B, H, sc + 3 * B, sc + 4 * B, R, D.ld, B, post, D.pcomp,
                               m->stream));
    if (m->sp) {
        double* scr = nullptr;
        if (int rc = mcd_sparse_scratch_(m->sp_handle, m->stream, B, &scr)) return rc;
        MHIP_TRY(mcd::launch_sparse_tree_logpdf(*m->sp, *m->sp_tree, H, R, D.ld, sc + 2 * B, sc + 3 * B, B, post + B, post + 2 * B, scr, m->stream));
        return MCD_OK;
    }
    MHIP_TRY(mcd::launch_tree_logpdf(*m->mvn, *m->tree, H, R, D.ld, sc + 2 * B, sc + 3 * B, B, post + B, post + 2 * B, m->stream));
    return MCD_OK;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// the part of mcd_mh_create / mcd_mh_create_sparse behind the handles: m->mvn / m->tree (dense) or m->sp / m->sp_tree / m->tree
// (= &m->tree_shim, sparse) and m->prior are set, `parent` is the host copy of the topology, host_L the host factor (dense) or null
int mh_create_impl(mcd_mh_t** out, std::unique_ptr<mcd_mh>& m, const mcd_prior_t* prior, int dev_t, int dev_p, const int32_t* parent, const double* host_L,
                   int n_prop, const int32_t* kind, const int32_t* node, const int32_t* n1, const int32_t* n2, const int32_t* jac_root,
                   const int32_t* dim, const double* p0, const double* p1, int64_t batch, uint64_t seed)
{
    if (dev_t != dev_p) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: tree (device %d) and prior (device %d) live on different GPUs", dev_t, dev_p);
    const int n = m->tree->n_nodes;
    if (m->prior->n_nodes != n) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: tree has %d nodes, prior %d", n, m->prior->n_nodes);
    {   // tree and prior must describe the same topology: compare the prior's parent array (device) with the tree's
        std::vector<int32_t> pp(n);
        MHIP_TRY(hipSetDevice(dev_p));
        MHIP_TRY(hipMemcpy(pp.data(), m->prior->parent, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
        for (int v = 0; v < n; ++v)
            if (pp[v] != parent[v]) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: tree and prior have different topologies (node %d)", v);
    }
    std::vector<int32_t> size(n, 1);
    for (int v = n - 1; v > 0; --v) size[parent[v]] += size[v];
    // the braces live in the prior's tables (device memory): a host copy for the checks below
    const int nbr = m->prior->n_brace;
    std::vector<int32_t> host_brace_ptr(nbr + 1, 0), host_brace_nodes;
    if (nbr > 0) {
        int dev_prior = 0;
        const mcd::PriorDev* pd = nullptr;
        (void)mcd_prior_internal_(prior, &pd, &dev_prior);
        MHIP_TRY(hipSetDevice(dev_prior));
        MHIP_TRY(hipMemcpy(host_brace_ptr.data(), pd->brace_ptr, sizeof(int32_t) * (nbr + 1), hipMemcpyDeviceToHost));
        host_brace_nodes.resize(host_brace_ptr[nbr]);
        MHIP_TRY(hipMemcpy(host_brace_nodes.data(), pd->brace_nodes, sizeof(int32_t) * host_brace_nodes.size(), hipMemcpyDeviceToHost));
    }
    // proposal table checks: the reference raises `error` for a path to a leaf / an invalid path when the proposal is built
    const int root_right = 1 + size[1];
    for (int i = 0; i < n_prop; ++i) {
        const int k = kind[i], v = node[i];
        if (!(p0[i] > 0) || !std::isfinite(p0[i])) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: proposal %d: p0 must be positive", i);
        if (dim[i] < 1) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: proposal %d: dimension must be >= 1", i);
        switch (k) {
            case MCD_PROP_SCALE_SCALAR:
                if (v < 0 || v > 4) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: proposal %d: scalar index %d", i, v);
                break;
            case MCD_PROP_SCALE_NORM_TREE:
                if (v != 2 && v != 3) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: proposal %d: norm must be tH (2) or rMu (3)", i);
                break;
            case MCD_PROP_SLIDE_NODE:
            case MCD_PROP_SCALE_SUBTREE_TIME:
                if (v < 1 || v >= n) return mfail(MCD_ERR_INVALID_ARG, "slideNodeAtUltrametric: Path is invalid (proposal %d, node %d).", i, v);
                if (size[v] == 1) return mfail(MCD_ERR_INVALID_ARG, "slideNodeAtUltrametric: Path leads to a leaf (proposal %d, node %d).", i, v);
                break;
            case MCD_PROP_PULLEY:
                if (size[1] == 1) return mfail(MCD_ERR_INVALID_ARG, "pulleyUltrametric: Left sub tree is a leaf.");
                if (size[root_right] == 1) return mfail(MCD_ERR_INVALID_ARG, "pulleyUltrametric: Right sub tree is a leaf.");
                break;
            case MCD_PROP_SCALE_BRANCH_RATE:
            case MCD_PROP_SCALE_SUBTREE_RATE:
                if (v < 1 || v >= n) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: proposal %d: node %d out of range", i, v);
                break;
            case MCD_PROP_SLIDE_NODE_CONTRA:
            case MCD_PROP_SCALE_SUBTREE_CONTRA:
                if (v < 1 || v >= n) return mfail(MCD_ERR_INVALID_ARG, "slideNodesAtContrarily: Path is invalid (proposal %d, node %d).", i, v);
                if (size[v] == 1) return mfail(MCD_ERR_INVALID_ARG, "slideNodesAtContrarily: Path leads to a leaf (proposal %d, node %d).", i, v);
                break;
            case MCD_PROP_SLIDE_BRACE:
            case MCD_PROP_SLIDE_BRACE_CONTRA:
                if (v < 0 || v >= m->prior->n_brace) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: proposal %d: the prior has no brace %d", i, v);
                for (int j = host_brace_ptr[v]; j < host_brace_ptr[v + 1]; ++j) {
                    const int x = host_brace_nodes[j];
                    if (x == 0) return mfail(MCD_ERR_INVALID_ARG, "slideBracedNodesUltrametric: Braced root node (proposal %d).", i);
                    if (size[x] == 1) return mfail(MCD_ERR_INVALID_ARG, "slideBracedNodesUltrametric: Path of a node leads to a leaf (proposal %d).", i);
                }
                break;
            case MCD_PROP_SLIDE_ROOT_CONTRA:
            case MCD_PROP_SCALE_RATES_TREE_CONTRA:
            case MCD_PROP_SCALE_VAR_TREE:
            case MCD_PROP_SCALE_VAR_TREE_AUTO: break;
            case MCD_PROP_SCALE_CONTRARILY:
                if (!(p1[i] > 0)) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: proposal %d: p1 must be positive", i);
                break;
            default: return mfail(MCD_ERR_UNSUPPORTED, "mcd_mh_create: proposal %d: unknown kind %d", i, k);
        }
    }
    // Which proposals move only a few branch distances (k_mh_chain_big.hip evaluates those by columns of L^-1 instead of a sweep):
    // a distance changes where a node's height, its parent's height or its rate changes -- for the node kinds below the node
    // itself with its daughters, or its sub tree.  The root's two daughters share distance slot 0.  This is a performance hint
    // only: the kernel finds the moved distances from the data and is right for any number of them.
    // (sparse_inc: the same for the two-launch path's incremental evaluation, k_mh_inc.hip, whose list of moved distances is not bounded
    // by registers: up to kMhIncSlots columns of L^-1 still cost less than a likelihood launch.)
    std::vector<int32_t> sparse((size_t)n_prop, 0), sparse_inc((size_t)n_prop, 0);
    const int opt_slots = mcd::opt_get(mcd::OPT_MH_INC_SLOTS);      // (mcd_set_option "MCD_MH_INC_SLOTS", read here)
    // (where the segment kernel runs the sparse proposals, k_mh_segment.hip, many more: a likelihood launch there also costs the whole
    // state's way through memory twice; measured at 1025 nodes x 512 chains: 16 -> 22.5, 64 -> 14.2, 128 -> 12.7, 192 -> 12.5 us per lock step)
    mcd::MhDev probe{};
    probe.n_nodes = n;
    probe.batch = batch;
    const bool seg_capable = m->mvn != nullptr && mcd::mh_segment_available(probe, *m->mvn);
    // (over a sparse precision matrix, k_mh_segment_sparse.hip: a listed distance costs a row of the matrix, some 15 entries, where the
    // two launches of a dense proposal cost a full product -- whatever the list holds)
    const bool sseg_capable = m->sp != nullptr && mcd::mh_segment_sparse_available(probe, *m->sp);
    const int list_cap = sseg_capable ? mcd::mh_segment_sparse_list() : mcd::kMhSegList;
    const int inc_slots = opt_slots != mcd::MCD_OPT_UNSET ? std::max(1, std::min(list_cap, opt_slots)) : sseg_capable ? list_cap : seg_capable ? mcd::kMhSegSlots : mcd::kMhIncSlots;
    // a tree whose distance slots ALL fit the sparse segment kernel's list: every proposal of the cycle can run inside a segment
    m->list_all = sseg_capable && m->sp->n <= list_cap;
    for (int pass = 0; pass < 2; ++pass) {
        const int opt_ss = mcd::opt_get(mcd::OPT_MH_SPARSE_SLOTS);      // (the streaming chain kernel's threshold; tuning: mcd_set_option "MCD_MH_SPARSE_SLOTS")
        // (up to 258 nodes the chain's likelihood wave takes the columns four at a time beside the prior: 48; above, where the chain wave
        // itself fetches them two at a time -- the sweep-only builds of the larger trees, which otherwise run in segments --: 16)
        const int big_default = (m->mvn != nullptr && m->mvn->R <= 4) ? mcd::kMhSparseSlots : 16;
        const int limit = pass ? inc_slots : opt_ss != mcd::MCD_OPT_UNSET ? std::max(1, std::min(64, opt_ss)) : big_default;      // (64: the kernel's list, kMhbList)
        std::vector<std::vector<int>> kids((size_t)n);
        for (int v = 1; v < n; ++v) kids[(size_t)parent[v]].push_back(v);
        auto slots_of = [&](const std::vector<int>& nodes) {
            std::vector<int> sl;
            for (int w : nodes) {
                if (w == 0) continue;
                const int key = (parent[w] == 0) ? 1 : w;              // both root daughters -> one slot
                if (std::find(sl.begin(), sl.end(), key) == sl.end()) sl.push_back(key);
            }
            return (int)sl.size();
        };
        for (int i = 0; i < n_prop; ++i) {
            const int k = kind[i], v = node[i];
            std::vector<int> touched;
            bool known = true;
            switch (k) {
                case MCD_PROP_SLIDE_NODE:
                case MCD_PROP_SLIDE_NODE_CONTRA:
                    touched.push_back(v);
                    for (int c : kids[(size_t)v]) touched.push_back(c);
                    break;
                case MCD_PROP_SCALE_BRANCH_RATE: touched.push_back(v); break;
                case MCD_PROP_SCALE_SUBTREE_TIME:
                case MCD_PROP_SCALE_SUBTREE_RATE:
                case MCD_PROP_SCALE_SUBTREE_CONTRA:
                    if (size[v] > 2 * limit) { known = false; break; }
                    for (int w = v; w < v + size[v]; ++w) touched.push_back(w);
                    break;
                case MCD_PROP_SLIDE_BRACE:
                case MCD_PROP_SLIDE_BRACE_CONTRA:
                    for (int j = host_brace_ptr[v]; j < host_brace_ptr[v + 1]; ++j) {
                        touched.push_back(host_brace_nodes[j]);
                        for (int c : kids[(size_t)host_brace_nodes[j]]) touched.push_back(c);
                    }
                    break;
                default: known = false; break;                         // scalars, whole-tree scalings, pulley, root slide
            }
            (pass ? sparse_inc : sparse)[(size_t)i] = ((known && slots_of(touched) <= limit) || (pass && m->list_all)) ? 1 : 0;
        }
    }
    m->sparse_rows = sparse_inc;
    m->device = dev_t;
    m->seed = seed;
    for (int i = 0; i < n_prop; ++i) m->rows.push_back(mcd::MhRow{kind[i], node[i], n1[i], n2[i], jac_root[i], p0[i], p1[i]});
    MHIP_TRY(hipSetDevice(m->device));
    MHIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    mcd::MhDev& D = m->dev;
    D.n_nodes = n;
    D.n_prop = n_prop;
    D.batch = batch;
    D.ld = (n + 7) / 8 * 8;
    D.chain0 = 0;
    D.parent = m->mvn ? m->tree->parent : m->prior->parent;
    D.brace_ptr = m->prior->brace_ptr;
    D.brace_nodes = m->prior->brace_nodes;
    D.n_brace = nbr;
    int rc = MCD_OK;
    const size_t B = (size_t)batch, BL = B * (size_t)D.ld, BP = B * (size_t)n_prop, BN = B * (size_t)n;
    if ((rc = dev_upload(m.get(), &D.size, size.data(), (size_t)n)) || (rc = dev_upload(m.get(), &D.kind, kind, (size_t)n_prop)) ||
        (rc = dev_upload(m.get(), &D.node, node, (size_t)n_prop)) || (rc = dev_upload(m.get(), &D.n1, n1, (size_t)n_prop)) ||
        (rc = dev_upload(m.get(), &D.n2, n2, (size_t)n_prop)) || (rc = dev_upload(m.get(), &D.jac_root, jac_root, (size_t)n_prop)) ||
        (rc = dev_upload(m.get(), &D.dim, dim, (size_t)n_prop)) || (rc = dev_upload(m.get(), &D.p0, p0, (size_t)n_prop)) ||
        (rc = dev_upload(m.get(), &D.p1, p1, (size_t)n_prop)) || (rc = dev_alloc(m.get(), &D.sc, 5 * B, true)) ||
        (rc = dev_alloc(m.get(), &D.H, BL, true)) || (rc = dev_alloc(m.get(), &D.R, BL, true)) ||
        (rc = dev_alloc(m.get(), &D.sc1, 5 * B, true)) || (rc = dev_alloc(m.get(), &D.H1, BL, true)) ||
        (rc = dev_alloc(m.get(), &D.R1, BL, true)) || (rc = dev_alloc(m.get(), &D.post, 3 * B, true)) ||
        (rc = dev_alloc(m.get(), &D.post1, 3 * B, true)) || (rc = dev_alloc(m.get(), &D.lnqj, B, true)) || (rc = dev_alloc(m.get(), &D.beta, B, false)) ||
        (rc = dev_alloc(m.get(), &D.tune, BP, false)) || (rc = dev_alloc(m.get(), &D.acc, BP, true)) ||
        (rc = dev_alloc(m.get(), &D.tried, BP, true)) || (rc = dev_alloc(m.get(), &D.age_sum, BN, true)) ||
        (rc = dev_alloc(m.get(), &D.age_sq, BN, true)) || (rc = dev_alloc(m.get(), &D.pcomp, 3 * B, true)) ||
        (rc = dev_alloc(m.get(), &D.pcomp1, 3 * B, true)) || (rc = dev_alloc(m.get(), &D.draws, 64 * 5 * B, true)) ||
        (rc = dev_alloc(m.get(), &D.pflags, B, true)) || (rc = dev_upload(m.get(), &D.sparse, sparse.data(), (size_t)n_prop)))
        return rc;
    // trees of at most 64 nodes: the whole schedule runs in one launch with the factor staged in LDS (k_mh_chain.hip).
    // MCD_MH_PER_PHASE=1 (diagnostic) keeps the two-launches-per-step path that larger trees use.
    const int nd = m->mvn ? m->mvn->n : m->sp->n;
    if (m->mvn && n <= 64 && !mcd::opt_is(mcd::OPT_MH_PER_PHASE, 1) && mcd::mh_chain_lds_bytes(nd, n_prop, 4) + sizeof(double) * mcd::prior_node_tables_doubles(m->prior->n_cal, m->prior->n_con) <= 64 * 1024) {
        std::vector<double> Fp((size_t)nd * 64, 0.0);
        for (int i = 0; i < nd; ++i) {
            const double inv = 1.0 / host_L[(size_t)i * nd + i];
            for (int j = 0; j < i; ++j) Fp[(size_t)j * 64 + i] = host_L[(size_t)i * nd + j] * inv;
        }
        if ((rc = dev_upload(m.get(), &m->d_Fp, Fp.data(), Fp.size()))) return rc;
        m->chain_kernel = true;
    }
    {
        std::vector<double> ones(BP > B ? BP : B, 1.0);
        MHIP_TRY(hipMemcpy(D.tune, ones.data(), sizeof(double) * BP, hipMemcpyHostToDevice));
        MHIP_TRY(hipMemcpy(D.beta, ones.data(), sizeof(double) * B, hipMemcpyHostToDevice));
    }
    *out = m.release();
    return MCD_OK;
}

}  // namespace

extern "C" {

int mcd_mh_create(mcd_mh_t** out, const mcd_tree_t* tree, const mcd_prior_t* prior, int n_prop, const int32_t* kind,
                  const int32_t* node, const int32_t* n1, const int32_t* n2, const int32_t* jac_root, const int32_t* dim,
                  const double* p0, const double* p1, int64_t batch, uint64_t seed)
{
    if (!out) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: out is NULL");
    *out = nullptr;
    if (!tree || !prior) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: NULL tree or prior handle");
    if (n_prop <= 0 || !kind || !node || !n1 || !n2 || !jac_root || !dim || !p0 || !p1)
        return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: empty or NULL proposal table");
    if (batch <= 0 || batch > (int64_t)1 << 31) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: batch must be in [1, 2^31]");
    std::unique_ptr<mcd_mh> m(new mcd_mh());
    int dev_t = 0, dev_p = 0;
    const int32_t* parent = nullptr;
    const double* host_L = nullptr;
    if (mcd_tree_internal_(tree, &m->mvn, &m->tree, &dev_t, &parent, &host_L) || mcd_prior_internal_(prior, &m->prior, &dev_p))
        return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create: invalid handle");
    return mh_create_impl(out, m, prior, dev_t, dev_p, parent, host_L, n_prop, kind, node, n1, n2, jac_root, dim, p0, p1, batch, seed);
}

// The same driver over a likelihood whose precision matrix stays sparse on the device (mcd_sparse_*): trees beyond the dense kernels'
// 1024 branches.  Two launches per lock step -- the workgroup-per-chain step kernel leaving the proposed distances, the sparse
// product on them (k_sparse.hip).
int mcd_mh_create_sparse(mcd_mh_t** out, const mcd_sparse_tree_t* tree, const mcd_prior_t* prior, int n_prop, const int32_t* kind,
                         const int32_t* node, const int32_t* n1, const int32_t* n2, const int32_t* jac_root, const int32_t* dim,
                         const double* p0, const double* p1, int64_t batch, uint64_t seed)
{
    if (!out) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create_sparse: out is NULL");
    *out = nullptr;
    if (!tree || !prior) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create_sparse: NULL tree or prior handle");
    if (n_prop <= 0 || !kind || !node || !n1 || !n2 || !jac_root || !dim || !p0 || !p1)
        return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create_sparse: empty or NULL proposal table");
    if (batch <= 0 || batch > (int64_t)1 << 31) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create_sparse: batch must be in [1, 2^31]");
    std::unique_ptr<mcd_mh> m(new mcd_mh());
    int dev_t = 0, dev_p = 0;
    const int32_t* parent = nullptr;
    if (mcd_sparse_tree_internal_(tree, &m->sp_handle, &m->sp, &m->sp_tree, &dev_t, &parent) || mcd_prior_internal_(prior, &m->prior, &dev_p))
        return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_create_sparse: invalid handle");
    if (m->sp_tree->n_nodes < 3 || m->sp_tree->n_nodes > 2048)
        return mfail(MCD_ERR_UNSUPPORTED, "mcd_mh_create_sparse: %d nodes (the sparse driver serves trees of 3 .. 2048 nodes)", m->sp_tree->n_nodes);
    m->tree_shim = mcd::TreeDev{m->sp_tree->n_nodes, (m->sp_tree->n_nodes + 63) / 64 * 64, m->sp_tree->root_right, m->prior->parent, m->sp_tree->slot_node,
                                m->sp_tree->slot_parent, nullptr, nullptr};
    m->tree = &m->tree_shim;
    return mh_create_impl(out, m, prior, dev_t, dev_p, parent, nullptr, n_prop, kind, node, n1, n2, jac_root, dim, p0, p1, batch, seed);
}

void mcd_mh_destroy(mcd_mh_t* m) { delete m; }

int mcd_mh_set_chain_offset(mcd_mh_t* m, int64_t first_chain)
{
    if (!m || first_chain < 0 || first_chain + m->dev.batch > ((int64_t)1 << 32))
        return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_set_chain_offset: chain indices must stay below 2^32");
    m->dev.chain0 = first_chain;
    return MCD_OK;
}

int mcd_mh_set_state(mcd_mh_t* m, const double* birth, const double* death, const double* tH, const double* heights,
                     const double* rMu, const double* rVar, const double* rates, int64_t ld_state)
{
    if (!m || !birth || !death || !tH || !heights || !rMu || !rVar || !rates) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_set_state: NULL argument");
    mcd::MhDev& D = m->dev;
    if (ld_state < D.n_nodes) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_set_state: ld_state < n_nodes");
    MHIP_TRY(hipSetDevice(m->device));
    const size_t B = (size_t)D.batch;
    const double* sc_src[5] = {birth, death, tH, rMu, rVar};
    for (int i = 0; i < 5; ++i) MHIP_TRY(hipMemcpyAsync(D.sc + i * B, sc_src[i], sizeof(double) * B, hipMemcpyHostToDevice, m->stream));
    MHIP_TRY(hipMemcpy2DAsync(D.H, sizeof(double) * D.ld, heights, sizeof(double) * ld_state, sizeof(double) * D.n_nodes, B, hipMemcpyHostToDevice, m->stream));
    MHIP_TRY(hipMemcpy2DAsync(D.R, sizeof(double) * D.ld, rates, sizeof(double) * ld_state, sizeof(double) * D.n_nodes, B, hipMemcpyHostToDevice, m->stream));
    if (int rc = eval_posterior(m, D.sc, D.H, D.R, D.post)) return rc;
    MHIP_TRY(hipStreamSynchronize(m->stream));
    m->have_state = true;
    return MCD_OK;
}

int mcd_mh_get_state(const mcd_mh_t* cm, double* birth, double* death, double* tH, double* heights, double* rMu, double* rVar,
                     double* rates, int64_t ld_state)
{
    if (!cm || !birth || !death || !tH || !heights || !rMu || !rVar || !rates) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_get_state: NULL argument");
    const mcd::MhDev& D = cm->dev;
    if (ld_state < D.n_nodes) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_get_state: ld_state < n_nodes");
    MHIP_TRY(hipSetDevice(cm->device));
    MHIP_TRY(hipStreamSynchronize(cm->stream));
    const size_t B = (size_t)D.batch;
    double* sc_dst[5] = {birth, death, tH, rMu, rVar};
    for (int i = 0; i < 5; ++i) MHIP_TRY(hipMemcpy(sc_dst[i], D.sc + i * B, sizeof(double) * B, hipMemcpyDeviceToHost));
    MHIP_TRY(hipMemcpy2D(heights, sizeof(double) * ld_state, D.H, sizeof(double) * D.ld, sizeof(double) * D.n_nodes, B, hipMemcpyDeviceToHost));
    MHIP_TRY(hipMemcpy2D(rates, sizeof(double) * ld_state, D.R, sizeof(double) * D.ld, sizeof(double) * D.n_nodes, B, hipMemcpyDeviceToHost));
    return MCD_OK;
}

int mcd_mh_get_posterior(const mcd_mh_t* cm, double* post)
{
    if (!cm || !post) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_get_posterior: NULL argument");
    const mcd::MhDev& D = cm->dev;
    MHIP_TRY(hipSetDevice(cm->device));
    MHIP_TRY(hipStreamSynchronize(cm->stream));
    const size_t B = (size_t)D.batch;
    std::vector<double> tmp(3 * B);
    MHIP_TRY(hipMemcpy(tmp.data(), D.post, sizeof(double) * 3 * B, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < B; ++b)
        for (int i = 0; i < 3; ++i) post[b * 3 + i] = tmp[i * B + b];
    return MCD_OK;
}

int mcd_mh_posterior_device(const mcd_mh_t* cm, const double** post, void** stream)
{
    if (!cm || !post) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_posterior_device: NULL argument");
    *post = cm->dev.post;                      // [3][batch]: ln prior, ln likelihood, ln jacobianRootBranch of the current states
    if (stream) *stream = (void*)cm->stream;   // the stream the sampler's launches are ordered on
    return MCD_OK;
}

int mcd_mh_last_path(const mcd_mh_t* m) { return m ? m->last_path : mfail(MCD_ERR_INVALID_ARG, "mcd_mh_last_path: NULL handle"); }
int64_t mcd_mh_last_dynamic_lds(const mcd_mh_t* m) { return m ? (int64_t)m->last_lds : (int64_t)mfail(MCD_ERR_INVALID_ARG, "mcd_mh_last_dynamic_lds: NULL handle"); }

// ---- Metropolis-coupled MCMC: the swap phase (k_mc3.hip) -----------------------------------------------------------------
int mcd_mh_mc3_init(mcd_mh_t* m, int n_chains, const double* betas, int64_t total_chains, uint64_t seed)
{
    if (!m || !betas) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_init: NULL argument");
    const mcd::MhDev& D = m->dev;
    if (n_chains < 2 || n_chains > 16) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_init: n_chains must be 2 .. 16");
    if (total_chains <= 0 || total_chains % n_chains != 0)
        return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_init: the global number of chains must be a multiple of n_chains");
    if (D.chain0 < 0 || D.chain0 + D.batch > total_chains) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_init: this handle's chains [%lld, %lld) lie outside 0 .. total_chains", (long long)D.chain0, (long long)(D.chain0 + D.batch));
    if (betas[0] != 1.0) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_init: betas[0] must be 1 (the cold chain)");
    for (int i = 1; i < n_chains; ++i)
        if (!(betas[i] > 0) || !(betas[i] < betas[i - 1])) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_init: betas must decrease and stay positive");
    MHIP_TRY(hipSetDevice(m->device));
    MHIP_TRY(hipStreamSynchronize(m->stream));
    mcd::Mc3Dev& C = m->mc3;
    double* ladder = nullptr;
    if (int rc = dev_alloc(m, &ladder, (size_t)n_chains, false)) return rc;
    if (int rc = dev_alloc(m, &C.rank, (size_t)total_chains, false)) return rc;
    if (int rc = dev_alloc(m, &C.tried, (size_t)n_chains, true)) return rc;
    if (int rc = dev_alloc(m, &C.accepted, (size_t)n_chains, true)) return rc;
    C.ladder = ladder;
    C.n_chains = n_chains;
    C.total = total_chains;
    std::vector<int32_t> rank((size_t)total_chains);
    std::vector<double> beta((size_t)D.batch);
    for (int64_t c = 0; c < total_chains; ++c) rank[(size_t)c] = (int32_t)(c % n_chains);
    for (int64_t b = 0; b < D.batch; ++b) beta[(size_t)b] = betas[(D.chain0 + b) % n_chains];
    MHIP_TRY(hipMemcpy(ladder, betas, sizeof(double) * (size_t)n_chains, hipMemcpyHostToDevice));
    MHIP_TRY(hipMemcpy(C.rank, rank.data(), sizeof(int32_t) * rank.size(), hipMemcpyHostToDevice));
    MHIP_TRY(hipMemcpy(D.beta, beta.data(), sizeof(double) * beta.size(), hipMemcpyHostToDevice));
    m->mc3_seed = seed;
    m->mc3_phase = 0;
    return MCD_OK;
}

int mcd_mh_mc3_swap(mcd_mh_t* m, int n_swaps, const double* gathered, int world, int64_t chains_per_rank)
{
    if (!m) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_swap: NULL handle");
    const mcd::MhDev& D = m->dev;
    const mcd::Mc3Dev& C = m->mc3;
    if (C.n_chains == 0) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_swap: call mcd_mh_mc3_init first");
    if (n_swaps < 1 || n_swaps > C.n_chains - 1) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_swap: need 1 <= n_swaps <= n_chains - 1");
    if (gathered == nullptr) {                           // one rank: the sampler's own [3][batch]
        if (C.total != D.batch || D.chain0 != 0) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_swap: a handle that holds a shard of the chains needs the gathered ln posteriors");
        gathered = D.post;
        world = 1;
        chains_per_rank = D.batch;
    } else {
        // the swap kernel reads global chain c at gathered[c / chains_per_rank][.][c % chains_per_rank] and writes this handle's temperatures at
        // c - chain0: equal shards, this handle holding exactly the shard of rank chain0 / chains_per_rank -- anything else would silently
        // read another chain's ln posterior
        if (world < 1 || chains_per_rank < 1 || (int64_t)world * chains_per_rank != C.total)
            return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_swap: world x chains_per_rank must equal the global number of chains (%lld)", (long long)C.total);
        if (chains_per_rank != D.batch)
            return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_swap: chains_per_rank (%lld) is not this handle's batch (%lld): the shards must be equally large",
                         (long long)chains_per_rank, (long long)D.batch);
        if (D.chain0 % chains_per_rank != 0 || D.chain0 / chains_per_rank >= world)
            return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_swap: this handle's first chain (%lld) is not the start of a shard of %lld chains", (long long)D.chain0,
                         (long long)chains_per_rank);
    }
    MHIP_TRY(hipSetDevice(m->device));
    // enqueued on the sampler's stream, behind the run and the all-gather that produced `gathered`: no host synchronisation
    MHIP_TRY(mcd::launch_mc3_swap(C, gathered, world, chains_per_rank, n_swaps, m->mc3_seed, m->mc3_phase, D.beta, D.chain0, D.batch, m->stream));
    m->mc3_phase += 1;
    return MCD_OK;
}

int mcd_mh_mc3_get(const mcd_mh_t* cm, int32_t* rank, int64_t* tried, int64_t* accepted, double* beta)
{
    if (!cm) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_get: NULL handle");
    const mcd::Mc3Dev& C = cm->mc3;
    if (C.n_chains == 0) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_mc3_get: call mcd_mh_mc3_init first");
    MHIP_TRY(hipSetDevice(cm->device));
    MHIP_TRY(hipStreamSynchronize(cm->stream));
    if (rank) MHIP_TRY(hipMemcpy(rank, C.rank, sizeof(int32_t) * (size_t)C.total, hipMemcpyDeviceToHost));
    static_assert(sizeof(unsigned long long) == sizeof(int64_t), "");
    if (tried) MHIP_TRY(hipMemcpy(tried, C.tried, sizeof(int64_t) * (size_t)(C.n_chains - 1), hipMemcpyDeviceToHost));
    if (accepted) MHIP_TRY(hipMemcpy(accepted, C.accepted, sizeof(int64_t) * (size_t)(C.n_chains - 1), hipMemcpyDeviceToHost));
    if (beta) MHIP_TRY(hipMemcpy(beta, cm->dev.beta, sizeof(double) * (size_t)cm->dev.batch, hipMemcpyDeviceToHost));
    return MCD_OK;
}

int mcd_mh_run(mcd_mh_t* m, const int32_t* schedule, int64_t n_iter, int32_t S, int accumulate, double* trace_alpha,
               int8_t* trace_accept)
{
    if (!m) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_run: NULL handle");
    if (!m->have_state) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_run: call mcd_mh_set_state first");
    if (n_iter < 0 || S <= 0 || (n_iter > 0 && !schedule)) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_run: bad schedule");
    if (n_iter == 0) return MCD_OK;
    mcd::note_dynamic_lds(0);
    mcd::MhDev& D = m->dev;
    const size_t steps = (size_t)n_iter * (size_t)S;
    for (size_t i = 0; i < steps; ++i)
        if (schedule[i] < 0 || schedule[i] >= D.n_prop) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_run: schedule[%zu] = %d is not a proposal row", i, schedule[i]);
    MHIP_TRY(hipSetDevice(m->device));
    if (steps > m->sched_cap) {
        if (m->d_sched) (void)hipFree(m->d_sched);
        m->d_sched = nullptr;
        m->sched_cap = 0;
        MHIP_TRY(hipMalloc((void**)&m->d_sched, sizeof(int32_t) * steps));
        m->sched_cap = steps;
    }
    MHIP_TRY(hipMemcpyAsync(m->d_sched, schedule, sizeof(int32_t) * steps, hipMemcpyHostToDevice, m->stream));
    const bool trace = trace_alpha || trace_accept;
    const size_t B = (size_t)D.batch;
    if (trace && steps * B > m->trace_cap) {
        if (m->d_trace_alpha) (void)hipFree(m->d_trace_alpha);
        if (m->d_trace_accept) (void)hipFree(m->d_trace_accept);
        m->d_trace_alpha = nullptr;
        m->d_trace_accept = nullptr;
        m->trace_cap = 0;
        MHIP_TRY(hipMalloc((void**)&m->d_trace_alpha, sizeof(double) * steps * B));
        MHIP_TRY(hipMalloc((void**)&m->d_trace_accept, steps * B));
        m->trace_cap = steps * B;
    }
    if (m->chain_kernel) {
        m->last_path = MCD_MH_PATH_CHAIN_LDS;
        MHIP_TRY(mcd::launch_mh_chain(D, *m->mvn, *m->tree, *m->prior, m->d_Fp, m->d_sched, (int64_t)steps, S, accumulate, m->step, m->seed,
                                      trace ? m->d_trace_alpha : nullptr, trace ? m->d_trace_accept : nullptr, m->stream));
        m->step += steps;
        if (accumulate) m->n_samples += n_iter;
    }
    // larger trees at a sampler's batch: the whole schedule in one launch as well, the factor streamed once per step
    // (k_mh_chain_big.hip).  MCD_MH_PER_PHASE=1 keeps the two-launch path (tests, timing; read per call).
    const bool per_phase = mcd::opt_is(mcd::OPT_MH_PER_PHASE, 1);
    const bool seg_off = mcd::opt_is(mcd::OPT_MH_SEGMENTS, 0), inc_off = mcd::opt_is(mcd::OPT_MH_INCREMENTAL, 0);
    // From 259 nodes (R >= 6) the segment path below is ahead of it (271 nodes x 512 chains 11.3 -> 9.6 us per lock step, 513 nodes
    // 16.0 -> 9.6): the factor streamed for every dense proposal costs more there than two launches for the few proposals that
    // move more than 192 distances.  MCD_MH_SEGMENTS=0 / MCD_MH_INCREMENTAL=0 keep the streaming kernel.
    const bool prefer_segments = m->mvn && !m->chain_kernel && !seg_off && !inc_off &&
                                 mcd::mh_segment_available(D, *m->mvn) && mcd::use_split(*m->mvn, std::min<int64_t>(D.batch, mcd::kSplitMaxBatch));
    const bool streaming = m->mvn && !m->chain_kernel && !per_phase && mcd::effective_form(*m->mvn) != MCD_FORM_MULTIPLY &&
                           mcd::mh_chain_big_available(D, *m->mvn) && !prefer_segments;
    if (streaming) {
        m->last_path = MCD_MH_PATH_CHAIN_STREAMED;
        // (launches of at most ~64 k steps, whole iterations each: a second or so of kernel time; what a launch costs -- the chains'
        // state in, out again -- is some 20 us)
        const int64_t per_launch = (int64_t)S * (65536 / S > 0 ? 65536 / S : 1);
        for (int64_t done = 0; done < (int64_t)steps; done += per_launch) {
            const int64_t now = ((int64_t)steps - done < per_launch) ? (int64_t)steps - done : per_launch;
            MHIP_TRY(mcd::launch_mh_chain_big(D, *m->mvn, *m->tree, *m->prior, m->d_sched + done, now, S, accumulate, m->step, m->seed,
                                              trace ? m->d_trace_alpha + done * B : nullptr, trace ? m->d_trace_accept + done * B : nullptr,
                                              m->stream));
            m->step += (uint64_t)now;
        }
        if (accumulate) m->n_samples += n_iter;
    }
    if (!m->chain_kernel && !streaming) {
        // two launches per step: [accept step s-1 + propose step s + ln prior] and [likelihood + root-branch Jacobian]
        const int64_t total = (int64_t)steps;
        if ((size_t)D.n_nodes * 32 > 64 * 1024) return mfail(MCD_ERR_UNSUPPORTED, "mcd_mh_run: more than 2048 nodes");
        const mcd::MhRow none{0, 0, 0, 0, 0, 1.0, 0.0};
        // the draws of schedule positions [64 k, 64 k + 64) are computed when position 64 k is about to be proposed
        const uint64_t step_base = m->step;                      // the step number of schedule position 0
        auto draws_for = [&](int64_t idx) -> int {
            if ((idx & 63) == 0) {
                const int count = (int)((total - idx < 64) ? total - idx : 64);
                const hipError_t e = mcd::launch_mh_draws(D, m->d_sched, idx, count, step_base + (uint64_t)idx, m->seed, m->stream);
                if (e != hipSuccess) return mfail(MCD_ERR_HIP, "launch_mh_draws: %s", hipGetErrorString(e));
            }
            return MCD_OK;
        };
        if (int rc = draws_for(0)) return rc;
        // The ln prior of a proposed state depends on the proposal only, like its ln likelihood: where the sweep serves the
        // likelihood launch, that launch carries the prior as workgroups of a second role (k_tree_logpdf.hip, mh_prior_role.hpp)
        // and k_mh_step leaves it out; elsewhere (row-split / multiply form) k_mh_step evaluates it as before.  Same functions
        // on the same numbers either way: the same chains.  MCD_MH_PRIOR=0 keeps it inside the step everywhere (tests, timing).
        const bool prior_in_step = mcd::opt_is(mcd::OPT_MH_PRIOR, 0);
        // (A launch of its own for the prior with four waves per chain was measured for the larger trees: 58.9 -> 56.4 us per
        // lock step at 1025 nodes, 33.3 -> 35.4 at 513 -- the step kernel's other strided loops weigh more there; not kept.)
        const bool beside = m->mvn && !prefer_segments && !prior_in_step && mcd::tree_logpdf_can_carry_prior(*m->mvn, D.batch, D.n_nodes);
        const int prior_inline = beside ? 0 : 1;
        // large trees: the step kernel (a workgroup per chain) leaves the proposed states' DISTANCES, the likelihood launch takes
        // them as plain vectors (the row-split kernel's tree staging costs 6 us more at 1023 slots); same arithmetic, same bits
        const int n_dim = m->mvn ? m->mvn->n : m->sp->n;
        const int wg_from = m->sp ? 0 : prefer_segments ? 258 : 320;   // (the workgroup-per-chain step kernel: the only one that leaves distances)
        const bool use_x = mcd::mh_step_wg_active(D, prior_inline, wg_from) && !beside && D.n_nodes > wg_from;
        if (m->sp && !use_x) return mfail(MCD_ERR_UNSUPPORTED, "mcd_mh_run: the sparse driver needs the workgroup-per-chain step kernel (MCD_MH_STEP_WG must not be 0)");
        if (use_x && m->d_X1 == nullptr) {
            MHIP_TRY(hipMalloc((void**)&m->d_X1, sizeof(double) * (size_t)D.batch * (size_t)n_dim));
            m->allocs.push_back(m->d_X1);
        }
        // the workgroup-per-chain step kernel keeps the per-node summands of the ln prior between launches (k_mh.hip: psum);
        // MCD_MH_PRIOR_CACHE=0: every summand at every step
        {
            const bool keep = prior_inline && mcd::mh_step_wg_active(D, prior_inline, wg_from) && !mcd::opt_is(mcd::OPT_MH_PRIOR_CACHE, 0);
            if (keep && m->d_psum == nullptr) {
                const size_t NS = (size_t)((D.n_nodes - 1 + 63) / 64) * 64;
                MHIP_TRY(hipMalloc((void**)&m->d_psum, sizeof(double) * (size_t)D.batch * 4 * NS));
                m->allocs.push_back(m->d_psum);
                MHIP_TRY(hipMalloc((void**)&m->d_psel, sizeof(int32_t) * 2 * (size_t)D.batch));
                m->allocs.push_back(m->d_psel);
            }
            D.psum = keep ? m->d_psum : nullptr;
            D.psel = keep ? m->d_psel : nullptr;
        }
        m->last_path = use_x ? MCD_MH_PATH_STEP_WG_X : beside ? MCD_MH_PATH_TWO_LAUNCH_PRIOR_BESIDE : MCD_MH_PATH_TWO_LAUNCH;
        const mcd::TreeDev* Tx = use_x ? m->tree : nullptr;
        double* X1 = use_x ? m->d_X1 : nullptr;
        // Large trees at a sampler's batch: the likelihood launch only for the proposals that move many distances (k_mh_inc.hip);
        // the others are evaluated from columns of L^-1 on the kept z.  MCD_MH_INCREMENTAL=0: the full evaluation at every step.
        // (batches beyond the row-split kernel's 1024 chains: its z products chunk by chunk, z' of a dense proposal copied to zprop)
        const bool inc_dense = m->mvn && use_x && !inc_off && m->mvn->Wc != nullptr && 64 * m->mvn->R <= 1024 &&
                               mcd::use_split(*m->mvn, std::min<int64_t>(D.batch, mcd::kSplitMaxBatch));
        // Over a sparse precision matrix (k_mh_segment_sparse.hip) the incremental form keeps the quadratic form q itself: MhInc with
        // NPz = 1 -- zcur / zprop = q of the current states / of the pending dense proposal -- so that k_mh_step_wg's accept half moves
        // X1 -> X0 and q' -> q like it moves z' -> z.  It exists only together with the segments.
        const bool inc_sparse = m->sp && use_x && !inc_off && !seg_off && mcd::mh_segment_sparse_available(D, *m->sp) && mcd::sparse_quad_available(*m->sp, D.batch);
        const bool inc = inc_dense || inc_sparse;
        const bool chunked = inc_dense && D.batch > mcd::kSplitMaxBatch;
        const int dense_mode = (chunked || inc_sparse) ? 1 : 2;   // where the z' (q') of a dense proposal is afterwards: zprop or the z tiles
        static const mcd::MvnDev no_mvn{};                   // (k_mh_step_wg takes the incremental bookkeeping only with an MvnDev beside it)
        const mcd::MvnDev* Vinc = m->mvn ? m->mvn : &no_mvn;
        mcd::MhInc& I = m->inc;
        if (inc_sparse && I.X0 == nullptr) {
            I.NPz = 1;
            MHIP_TRY(hipMalloc((void**)&I.X0, sizeof(double) * (size_t)D.batch * (size_t)n_dim));
            m->allocs.push_back(I.X0);
            MHIP_TRY(hipMalloc((void**)&I.zcur, sizeof(double) * 2 * (size_t)D.batch));
            m->allocs.push_back(I.zcur);
            I.zprop = I.zcur + D.batch;
        }
        if (inc_dense && I.X0 == nullptr) {
            I.NPz = 64 * m->mvn->R;
            MHIP_TRY(hipMalloc((void**)&I.X0, sizeof(double) * (size_t)D.batch * (size_t)n_dim));
            m->allocs.push_back(I.X0);
            MHIP_TRY(hipMalloc((void**)&I.zcur, sizeof(double) * (size_t)D.batch * (size_t)I.NPz));
            m->allocs.push_back(I.zcur);
            MHIP_TRY(hipMalloc((void**)&I.zprop, sizeof(double) * (size_t)D.batch * (size_t)I.NPz));
            m->allocs.push_back(I.zprop);
            MHIP_TRY(hipMalloc((void**)&m->d_inc_ll, sizeof(double) * (size_t)D.batch));
            m->allocs.push_back(m->d_inc_ll);
        }
        // ll and z = L^-1 (X - mu) of every chain by full products (the row-split kernel, at most 1024 chains per launch); z to dst
        // (chain-major) -- or, for a batch of one launch and dst = null, left in that launch's z tiles
        auto z_product = [&](const double* X, double* ll, double* dst) -> int {
            if (inc_sparse) {                                // the full form in one launch (k_sparse.hip: k_sparse_quad): ll and q (dst)
                MHIP_TRY(mcd::launch_sparse_quad(*m->sp, nullptr, X, nullptr, n_dim, nullptr, nullptr, D.batch, ll, nullptr, dst, m->stream));
                return MCD_OK;
            }
            for (int64_t c0 = 0; c0 < D.batch; c0 += mcd::kSplitMaxBatch) {
                const int64_t cnt = std::min<int64_t>(mcd::kSplitMaxBatch, D.batch - c0);
                MHIP_TRY(mcd::launch_logpdf_split_z(*m->mvn, X + c0 * n_dim, n_dim, cnt, ll + c0, &I.zt, &I.nr, m->stream));
                if (dst) MHIP_TRY(mcd::launch_mh_inc_take_z(I, dst, c0, cnt, m->stream));
            }
            return MCD_OK;
        };
        // (dense: the ll of that product is not used -- q is |z'|^2 afresh at every step; sparse: q itself is what is kept, and the chains' ln
        // likelihood is set to the recomputed value with it, so that the two stay the same number)
        auto refresh_z = [&]() -> int { return z_product(I.X0, inc_sparse ? D.post + D.batch : m->d_inc_ll, I.zcur); };
        if (inc) {
            I.mode = 0;
            MHIP_TRY(mcd::launch_mh_inc_init(D, *m->tree, I, n_dim, n_dim, m->stream));
            if (int rc = refresh_z()) return rc;
        }
        auto moves_likelihood = [&](int p) { return !(m->rows[p].kind == MCD_PROP_SCALE_SCALAR && (m->rows[p].node == 0 || m->rows[p].node == 1 || m->rows[p].node == 4)); };
        auto inc_mode = [&](int p) { return p < 0 ? 0 : !moves_likelihood(p) ? 0 : m->sparse_rows[(size_t)p] ? 1 : 2; };
        // Trees of 259 .. 1026 nodes: the runs of steps between two dense proposals as ONE launch each, every chain's state in LDS from
        // the run's first step to its last (k_mh_chain_big.hip, SEG); a dense proposal as before: proposed by the step kernel, its
        // likelihood by the row-split launch, accepted by the step kernel.  MCD_MH_SEGMENTS=0: every step by the two launches.
        const bool segments = inc_sparse || (inc_dense && !seg_off && mcd::mh_segment_available(D, *m->mvn) && I.NPz == 64 * m->mvn->R);
        if (segments) {
            bool summands_kept = false;                      // MhDev::psum holds the current states' summands
            int64_t draws_block = 0;                         // (draws_for(0) above)
            auto need_draws = [&](int64_t idx) -> int {
                if ((idx >> 6) != draws_block) {
                    draws_block = idx >> 6;
                    return draws_for(idx & ~(int64_t)63);
                }
                return MCD_OK;
            };
            // a dense proposal followed by a segment is decided by that segment's launch (k_mh_segment.hip: MhSegPending), not by a
            // launch of the step kernel that would do nothing else
            mcd::MhSegPending pending{};
            pending.p_acc = -1;
            bool have_pending = false;
            // ... and a dense proposal that follows a segment is PROPOSED by that segment's launch, from the state it holds in LDS
            // (MhSegPending::p_tail), not by a launch of the step kernel that reads everything back first.  MCD_MH_SEG_TAIL=0: by the step kernel.
            const bool tails = D.psum != nullptr && !mcd::opt_is(mcd::OPT_MH_SEG_TAIL, 0);
            bool proposed = false;                           // schedule[gs] is already proposed (by the segment before it)
            int64_t gs = 0;
            while (gs < total) {
                if (inc_mode(schedule[gs]) != 2) {
                    int64_t e = gs + 1;                      // ... up to the next recomputation of z (every 256 steps)
                    while (e < total && inc_mode(schedule[e]) != 2 && (e & 255) != 0) ++e;
                    if (!have_pending) pending.p_acc = -1;
                    pending.p_tail = (tails && e < total && inc_mode(schedule[e]) == 2) ? schedule[e] : -1;
                    pending.X1_tail = X1;
                    proposed = pending.p_tail >= 0;
                    if (inc_sparse)
                        MHIP_TRY(mcd::launch_mh_segment_sparse(D, *m->sp, *m->tree, *m->prior, I, m->d_sched + gs, e - gs, S, accumulate ? 1 : 0,
                                                               step_base + (uint64_t)gs, m->seed, trace ? m->d_trace_alpha + gs * B : nullptr,
                                                               trace ? m->d_trace_accept + gs * B : nullptr, gs, summands_kept ? 1 : 0, &pending,
                                                               m->list_all ? 1 : 0, m->stream));
                    else
                        MHIP_TRY(mcd::launch_mh_segment(D, *m->mvn, *m->tree, *m->prior, I, m->d_sched + gs, e - gs, S, accumulate ? 1 : 0, step_base + (uint64_t)gs,
                                                        m->seed, trace ? m->d_trace_alpha + gs * B : nullptr, trace ? m->d_trace_accept + gs * B : nullptr, gs,
                                                        summands_kept ? 1 : 0, &pending, m->stream));
                    have_pending = false;
                    if (D.psum != nullptr) summands_kept = true;
                    if (accumulate) m->n_samples += (e / S) - (gs / S);          // iterations closed by steps gs .. e - 1
                    m->step += (uint64_t)(e - gs);
                    gs = e;
                    if ((gs & 255) == 0 && gs < total)
                        if (int rc = refresh_z()) return rc;
                    continue;
                }
                // a dense proposal (and those that follow it directly)
                if (!proposed) {
                    if (int rc = need_draws(gs)) return rc;
                    I.mode = 0;
                    I.prop_mode = 2;
                    MHIP_TRY(mcd::launch_mh_step(D, *m->prior, -1, 0, schedule[gs], m->rows[schedule[gs]], (int)(gs & 63), m->step, m->seed, 0, nullptr,
                                                 nullptr, prior_inline, Tx, n_dim, X1, n_dim, m->stream, &I, Vinc, summands_kept ? 0 : 1));
                    if (D.psum != nullptr) summands_kept = true;
                }
                proposed = false;
                while (true) {
                    const int pa = schedule[gs];
                    if (int rc = z_product(X1, D.post1 + D.batch, (chunked || inc_sparse) ? I.zprop : nullptr)) return rc;
                    I.mode = dense_mode;
                    const bool closes = ((gs + 1) % S) == 0;
                    const bool refresh_now = ((gs + 1) & 255) == 0;
                    const int pn = (gs + 1 < total && inc_mode(schedule[gs + 1]) == 2) ? schedule[gs + 1] : -1;
                    if (pn >= 0)
                        if (int rc = need_draws(gs + 1)) return rc;
                    if (pn < 0 && gs + 1 < total && !refresh_now && D.psum != nullptr && summands_kept) {
                        pending.p_acc = pa;
                        pending.jac_root = m->rows[pa].jac_root;
                        pending.accumulate = (accumulate && closes) ? 1 : 0;
                        pending.step = m->step;
                        pending.trace_alpha = trace ? m->d_trace_alpha + gs * B : nullptr;
                        pending.trace_accept = trace ? m->d_trace_accept + gs * B : nullptr;
                        pending.X1 = X1;
                        pending.z_in_zprop = chunked ? 1 : 0;
                        have_pending = true;
                        m->step += 1;
                        if (accumulate && closes) m->n_samples += 1;
                        gs += 1;
                        break;
                    }
                    I.prop_mode = 2;
                    MHIP_TRY(mcd::launch_mh_step(D, *m->prior, pa, m->rows[pa].jac_root, pn, pn >= 0 ? m->rows[pn] : none, (int)((gs + 1) & 63), m->step,
                                                 m->seed, (accumulate && closes) ? 1 : 0, trace ? m->d_trace_alpha + gs * B : nullptr,
                                                 trace ? m->d_trace_accept + gs * B : nullptr, prior_inline, Tx, n_dim, X1, n_dim, m->stream, &I, Vinc, 0));
                    if (refresh_now && gs + 1 < total)
                        if (int rc = refresh_z()) return rc;
                    m->step += 1;
                    if (accumulate && closes) m->n_samples += 1;
                    gs += 1;
                    if (pn < 0) break;
                }
            }
            m->last_path = inc_sparse ? MCD_MH_PATH_SEGMENTS_SPARSE : MCD_MH_PATH_SEGMENTS;
        } else {
        I.prop_mode = inc ? inc_mode(schedule[0]) : 0;
        MHIP_TRY(mcd::launch_mh_step(D, *m->prior, -1, 0, schedule[0], m->rows[schedule[0]], 0, m->step - 1, m->seed, 0, nullptr, nullptr,
                                     prior_inline, Tx, n_dim, X1, n_dim, m->stream, inc ? &I : nullptr, m->mvn));
        for (int64_t gs = 0; gs < total; ++gs) {
            const int pa = schedule[gs];
            if (inc) {
                I.mode = inc_mode(pa);                       // the step kernel evaluated modes 0 and 1 itself
                if (I.mode == 2) {
                    if (int rc = z_product(X1, D.post1 + D.batch, chunked ? I.zprop : nullptr)) return rc;
                    I.mode = dense_mode;
                }
            } else if (m->sp) {
                double* scr = nullptr;
                if (int rc = mcd_sparse_scratch_(m->sp_handle, m->stream, D.batch, &scr)) return rc;
                MHIP_TRY(mcd::launch_sparse_logpdf(*m->sp, X1, n_dim, D.batch, D.post1 + D.batch, scr, m->stream));
            } else if (use_x)
                MHIP_TRY(mcd::launch_logpdf(*m->mvn, X1, n_dim, D.batch, D.post1 + D.batch, m->stream));
            else if (beside)
                MHIP_TRY(mcd::launch_tree_logpdf_with_prior(*m->mvn, *m->tree, D.H1, D.R1, D.ld, D.sc1 + 2 * D.batch, D.sc1 + 3 * D.batch, D.batch,
                                                            D.post1 + D.batch, D.post1 + 2 * D.batch, D, *m->prior, m->stream));
            else
                MHIP_TRY(mcd::launch_tree_logpdf(*m->mvn, *m->tree, D.H1, D.R1, D.ld, D.sc1 + 2 * D.batch, D.sc1 + 3 * D.batch, D.batch,
                                                 D.post1 + D.batch, D.post1 + 2 * D.batch, m->stream));
            const bool closes = ((gs + 1) % S) == 0;
            const int pn = (gs + 1 < total) ? schedule[gs + 1] : -1;
            if (pn >= 0)
                if (int rc = draws_for(gs + 1)) return rc;
            const bool refresh_now = inc && ((gs + 1) & 255) == 0;
            I.prop_mode = inc ? inc_mode(pn) : 0;
            MHIP_TRY(mcd::launch_mh_step(D, *m->prior, pa, m->rows[pa].jac_root, pn, pn >= 0 ? m->rows[pn] : none, (int)((gs + 1) & 63), m->step,
                                         m->seed, (accumulate && closes) ? 1 : 0, trace ? m->d_trace_alpha + gs * B : nullptr,
                                         trace ? m->d_trace_accept + gs * B : nullptr, prior_inline, Tx, n_dim, X1, n_dim, m->stream,
                                         inc ? &I : nullptr, m->mvn));
            if (refresh_now)                                 // (X0 is exact; z has been updated column by column since the last full product)
                if (int rc = refresh_z()) return rc;
            m->step += 1;
            if (accumulate && closes) m->n_samples += 1;
        }
        if (inc) m->last_path = MCD_MH_PATH_STEP_WG_INCREMENTAL;
        if (m->sp) m->last_path = MCD_MH_PATH_STEP_WG_SPARSE;
        }
    }
    m->last_lds = mcd::last_dynamic_lds();
    if (trace_alpha) MHIP_TRY(hipMemcpyAsync(trace_alpha, m->d_trace_alpha, sizeof(double) * steps * B, hipMemcpyDeviceToHost, m->stream));
    if (trace_accept) MHIP_TRY(hipMemcpyAsync(trace_accept, m->d_trace_accept, steps * B, hipMemcpyDeviceToHost, m->stream));
    MHIP_TRY(hipStreamSynchronize(m->stream));
    return MCD_OK;
}

int mcd_mh_tune(mcd_mh_t* m)
{
    if (!m) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_tune: NULL handle");
    MHIP_TRY(hipSetDevice(m->device));
    MHIP_TRY(mcd::launch_mh_tune(m->dev, m->stream));
    MHIP_TRY(hipStreamSynchronize(m->stream));
    return MCD_OK;
}

int mcd_mh_get_tuning(const mcd_mh_t* cm, double* tune, int32_t* accepted, int32_t* tried)
{
    if (!cm) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_get_tuning: NULL handle");
    const mcd::MhDev& D = cm->dev;
    MHIP_TRY(hipSetDevice(cm->device));
    MHIP_TRY(hipStreamSynchronize(cm->stream));
    const size_t BP = (size_t)D.batch * (size_t)D.n_prop;
    if (tune) MHIP_TRY(hipMemcpy(tune, D.tune, sizeof(double) * BP, hipMemcpyDeviceToHost));
    if (accepted) MHIP_TRY(hipMemcpy(accepted, D.acc, sizeof(int32_t) * BP, hipMemcpyDeviceToHost));
    if (tried) MHIP_TRY(hipMemcpy(tried, D.tried, sizeof(int32_t) * BP, hipMemcpyDeviceToHost));
    return MCD_OK;
}

int mcd_mh_set_tuning(mcd_mh_t* m, const double* tune)
{
    if (!m || !tune) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_set_tuning: NULL argument");
    const mcd::MhDev& D = m->dev;
    const size_t BP = (size_t)D.batch * (size_t)D.n_prop;
    for (size_t i = 0; i < BP; ++i)
        if (!(tune[i] > 0) || !std::isfinite(tune[i])) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_set_tuning: tuning parameters must be positive");
    MHIP_TRY(hipSetDevice(m->device));
    MHIP_TRY(hipStreamSynchronize(m->stream));
    MHIP_TRY(hipMemcpy(D.tune, tune, sizeof(double) * BP, hipMemcpyHostToDevice));
    return MCD_OK;
}

int mcd_mh_set_temperatures(mcd_mh_t* m, const double* beta)
{
    if (!m || !beta) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_set_temperatures: NULL argument");
    const mcd::MhDev& D = m->dev;
    for (int64_t b = 0; b < D.batch; ++b)
        if (!(beta[b] > 0) || !(beta[b] <= 1.0)) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_set_temperatures: reciprocal temperatures must be in (0, 1]");
    MHIP_TRY(hipSetDevice(m->device));
    MHIP_TRY(hipStreamSynchronize(m->stream));
    MHIP_TRY(hipMemcpy(D.beta, beta, sizeof(double) * (size_t)D.batch, hipMemcpyHostToDevice));
    return MCD_OK;
}

int mcd_mh_reset_counters(mcd_mh_t* m)
{
    if (!m) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_reset_counters: NULL handle");
    const mcd::MhDev& D = m->dev;
    const size_t BP = (size_t)D.batch * (size_t)D.n_prop;
    MHIP_TRY(hipSetDevice(m->device));
    MHIP_TRY(hipMemsetAsync(D.acc, 0, sizeof(int32_t) * BP, m->stream));
    MHIP_TRY(hipMemsetAsync(D.tried, 0, sizeof(int32_t) * BP, m->stream));
    MHIP_TRY(hipStreamSynchronize(m->stream));
    return MCD_OK;
}

int mcd_mh_get_age_sums(const mcd_mh_t* cm, double* age_sum, double* age_sq, int64_t* n_samples)
{
    if (!cm) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_get_age_sums: NULL handle");
    const mcd::MhDev& D = cm->dev;
    MHIP_TRY(hipSetDevice(cm->device));
    MHIP_TRY(hipStreamSynchronize(cm->stream));
    const size_t BN = (size_t)D.batch * (size_t)D.n_nodes;
    if (age_sum) MHIP_TRY(hipMemcpy(age_sum, D.age_sum, sizeof(double) * BN, hipMemcpyDeviceToHost));
    if (age_sq) MHIP_TRY(hipMemcpy(age_sq, D.age_sq, sizeof(double) * BN, hipMemcpyDeviceToHost));
    if (n_samples) *n_samples = cm->n_samples;
    return MCD_OK;
}

int mcd_mh_reset_age_sums(mcd_mh_t* m)
{
    if (!m) return mfail(MCD_ERR_INVALID_ARG, "mcd_mh_reset_age_sums: NULL handle");
    const mcd::MhDev& D = m->dev;
    const size_t BN = (size_t)D.batch * (size_t)D.n_nodes;
    MHIP_TRY(hipSetDevice(m->device));
    MHIP_TRY(hipMemsetAsync(D.age_sum, 0, sizeof(double) * BN, m->stream));
    MHIP_TRY(hipMemsetAsync(D.age_sq, 0, sizeof(double) * BN, m->stream));
    MHIP_TRY(hipStreamSynchronize(m->stream));
    m->n_samples = 0;
    return MCD_OK;
}

}  // extern "C"
