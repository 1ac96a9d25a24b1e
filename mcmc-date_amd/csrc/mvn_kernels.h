// mvn_kernels.h -- device-side operand descriptors and kernel launchers (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcd {

// Immutable device operands of one MVN (created once per analysis; app/Main.hs:333-347 builds the
// equivalent closure over mu, Sigma^-1, logdet in the reference).
struct MvnDev {
    int n;          // MVN dimension N (= 2 leaves - 3 for a tree)
    int R;          // 64-row blocks per chain held in registers; NP = 64 R >= N
    int ncols;      // columns swept (N rounded up to the chunk size of Cfg<R>)
    double c;       // -N * ln sqrt(2 pi)                     (app/Probability.hs:172-173)
    double logdet;  // log det Sigma
    const double* mu;       // [NP], zero padded
    const double* invdiag;  // [NP], 1 / L_ii, one padded
    const double* Ft;       // forward factor  L[i][j]/L[i][i], pair-interleaved column layout, NP*NP
    const double* Ut;       // backward factor L[i][r]/L[r][r], same layout, NP*NP
    const double* Wt;       // W = L^-1 as 16 x 4 MFMA operand tiles (host_factor.h: pack_w_tiles), for k_wide.hip
    const double* Wtb;      // the tiles of W^T for the gradient's second product (k_wide_grad.hip)
    const double* Wc;       // W = L^-1 column by column, [N][NP] (column j = NP doubles, zero above row j and in the padding): the
                            // incremental evaluation of sparse proposals in k_mh_chain_big.hip; NULL for R > 4
    const struct SplitHost* split;   // HOST pointer (never dereferenced on the device): schedules + scratch of the row-split form (k_split.hip)
    const int* form;        // HOST pointer: this handle's form override (MCD_FORM_*; 0 = the process default), mcd_mvn_set_form
};

// Row-split multiply form (k_split.hip).  One tile stream per (handle, G): the row blocks of W = L^-1 dealt to G row groups of
// 8 waves, every group's tiles packed in the order its waves consume them (host_factor.cpp: build_split_schedule; the
// schedule itself is arithmetic, split_sched.hpp).
constexpr int SP_MAXSEG = 10;                 // segments (runs of k tiles inside one row block) per wave, at most
struct SplitSched {
    int G;                  // row groups per 16-chain tile (0: this variant does not exist for the handle's N)
    int nc;                 // 256-column chunks of the residuals a workgroup stages (ceil(16 NB / 256))
    int NB;                 // row blocks of 16 rows
    const double* Ws;       // tile streams of all groups (device)
    int base[32];           // first tile of group g's stream
    int cuts[32][8];        // group g's row blocks that are cut between waves: [0] = how many, then wf | wl << 4 | block << 8
                            // (sp_for_each_cut, worked out on the host: hundreds of scalar instructions per workgroup otherwise)
};

// Topology tables of the time/rate trees (pre-order node ids, root = 0).
struct TreeDev {
    int n_nodes;
    int n_nodes_pad;
    int root_right;           // second child of the root (the first one is node 1)
    const int32_t* parent;    // [n_nodes]
    const int32_t* slot_node; // [NP] node whose branch feeds distance slot i (slot 0: node 1), -1 padded
    const int32_t* slot_parent; // [NP] parent of that node (0 padded): saves a dependent load where only the branch is needed
    const int32_t* child_ptr; // [n_nodes + 1] CSR children
    const int32_t* child_idx; // [n_nodes - 1]
};

// Everything `priorFunction ht md cb cs bs` closes over (app/Probability.hs:127-150), on pre-order node ids.
struct PriorDev {
    int n_nodes;
    int clock_model;          // 0 UncorrelatedGamma, 1 UncorrelatedLogNormal, 2 UncorrelatedWhiteNoise, 3 AutocorrelatedLogNormal
    double ht;                // initial, constant, approximate absolute time tree height
    const int32_t* parent;    // [n_nodes]
    const int32_t* first_child;   // [n_nodes], -1 for tips
    const int32_t* n_children;    // [n_nodes]
    const int32_t* second_child;  // [n_nodes], -1 when there is none
    int n_cal;
    const int32_t *cal_node, *cal_has_lo, *cal_has_hi;
    const double *cal_lo, *cal_lo_p, *cal_hi, *cal_hi_p;
    int n_con;
    const int32_t *con_young, *con_old;
    const double* con_p;
    int n_brace;
    const int32_t *brace_ptr, *brace_nodes;
    const double* brace_sd;
};

// doubles of LDS for the calibration and constraint tables of a persistent sampler kernel (prior_device.hpp: prior_stage_node_tables); 0 = none
__host__ __device__ inline size_t prior_node_tables_doubles(int n_cal, int n_con)
{
    if (n_cal + n_con == 0) return 0;
    return 4 * (size_t)n_cal + (size_t)n_con + (3 * (size_t)n_cal + 2 * (size_t)n_con + 1) / 2;
}

// Lock-step Metropolis-Hastings workspace of one batch of chains (k_mh.hip); all pointers are device memory.
struct MhDev {
    int n_nodes, n_prop;
    int64_t batch, ld;
    int64_t chain0;          // global index of chain 0 (random stream id = chain0 + b)
    const int32_t* parent;   // [n_nodes]
    const int32_t* size;     // [n_nodes] nodes in the sub tree of v (pre-order: the range [v, v + size[v]))
    const int32_t *kind, *node, *n1, *n2, *jac_root, *dim;   // [n_prop] proposal table (MCD_PROP_*)
    const double *p0, *p1;
    const int32_t *brace_ptr, *brace_nodes;                  // CSR of the braces (shared with the prior tables)
    int n_brace;
    double *sc, *H, *R;        // current state: scalars [5][batch] (birth, death, tH, rMu, rVar), heights/rates [batch][ld]
    double *sc1, *H1, *R1;     // proposed state
    double *post, *post1;      // [3][batch] ln prior, ln likelihood, ln jacobianRootBranch (current, proposed)
    double* lnqj;              // [batch] ln (q-ratio * Jacobian) of the pending proposal
    double* beta;              // [batch] reciprocal temperatures (1 = cold chain); prior and likelihood are raised to beta
    double* tune;              // [batch][n_prop]
    int32_t *acc, *tried;      // [batch][n_prop]
    double *age_sum, *age_sq;  // [batch][n_nodes]
    // two-launch path (trees with more than 64 nodes): the three blocks of the ln prior of the current / proposed state,
    // [batch][3], and the state-independent draws of a block of 64 steps, [64][batch][5] (k_mh_draws)
    double *pcomp, *pcomp1;
    double* draws;
    int32_t* pflags;           // [batch] which blocks of the ln prior the pending proposal moved (bit 0 nodes, 1 birth-death, 2 clock)
    const int32_t* sparse;     // [n_prop] 1: the proposal moves at most kMhSparseSlots distances (k_mh_chain_big: z updated by columns of L^-1)
    // k_mh_step_wg: the per-node summands of the birth-death and the clock block kept between launches, [batch][4][64 ceil((n_nodes - 1) / 64)]
    // (birth-death buffers 0 / 1, clock buffers 0 / 1); psel [batch][2]: which buffer of each block is the current state's (bit 0, bit 1), and
    // the blocks the pending proposal wrote (the bits of pflags); null: not kept
    double* psum;
    int32_t* psel;
};
constexpr int kMhSparseSlots = 48;      // (257 nodes x 512 chains, us per lock step: 8 -> 8.96, 32 -> 8.35 before the likelihood wave; with it 24 -> 7.70, 32 -> 7.52, 48 -> 7.44, 64 -> 7.44)
constexpr int kMhIncSlots = 32;       // the same bound for the two-launch path's incremental evaluation (k_mh_inc.hip; MCD_MH_INC_SLOTS)
constexpr int kMhSegSlots = 192;      // ... where the segment kernel runs the sparse proposals (k_mh_segment.hip)
constexpr int kMhSegList = 256;       // ... and what its list of moved distances holds: no proposal of a segment may move more

// The sparse form (k_sparse.hip): the precision matrix in CSR, any N up to kSparseMaxDim; all pointers are device memory.
struct SparseDev {
    int n;
    int64_t nnz;
    const int32_t* rowptr;   // [n + 1]
    const int32_t* trow;     // [nnz] row of every stored entry (the triplets are sorted by row, then column)
    const int32_t* col;      // [nnz], ascending inside a row
    const double* val;       // [nnz]
    const double* mu;        // [n]
    double c, logdet;        // -N ln sqrt(2 pi), log det Sigma
    // the one-launch quadratic form (k_sparse_quad): the entries as a flat stream, (row | column << 16, value); when the matrix is
    // exactly symmetric only its upper triangle, off-diagonal values doubled
    int64_t q_nnz;
    const uint32_t* q_rc;    // [q_nnz]
    const double* q_val;     // [q_nnz]
    // the symmetric part (P + P^T) / 2 in CSR -- the matrix itself when it is symmetric: the gradient -1/2 (P + P^T) dx and the
    // incremental form of the Metropolis-Hastings driver (k_mh_segment_sparse.hip: the rows of the moved distances)
    int64_t s_nnz;
    const int32_t* s_rowptr;   // [n + 1]
    const int32_t* s_trow;     // [s_nnz]
    const int32_t* s_col;      // [s_nnz]
    const double* s_val;       // [s_nnz]
    // ... and the same rows as fixed-size records (ELL, kSparseEllW entries per row, zero padded: no row pointer to chase -- a row's entries
    // are ONE round trip after its index is known): column, value and mu[column] side by side; ell_more[j] = entries of row j beyond the
    // record (they stay in the CSR arrays from s_rowptr[j] + kSparseEllW on).  The Metropolis-Hastings driver's incremental form.
    const int32_t* ell_col;    // [n][kSparseEllW]
    const double* ell_val;     // [n][kSparseEllW]
    const double* ell_mu;      // [n][kSparseEllW]
    const int32_t* ell_more;   // [n]
};
constexpr int kSparseEllW = 16;
struct SparseTreeDev {
    int n_nodes, root_right;
    const int32_t* slot_node;     // [n] distance slot -> node (getBranches . sumFirstTwo order)
    const int32_t* slot_parent;   // [n] that node's parent
};
constexpr int kSparseMaxDim = 8192;
// scratch: sparse_scratch_doubles(n, batch, gradient?) doubles of device memory that live until the launches have run
size_t sparse_scratch_doubles(int n, int64_t batch, bool grad);
hipError_t launch_sparse_logpdf(const SparseDev& S, const double* X, int64_t ldx, int64_t batch, double* ll, double* scratch, hipStream_t st);
hipError_t launch_sparse_grad(const SparseDev& S, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg, double* scratch,
                              hipStream_t st);
hipError_t launch_sparse_tree_logpdf(const SparseDev& S, const SparseTreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                     const double* rMu, int64_t batch, double* ll, double* logjac, double* scratch, hipStream_t st);
// ONE launch, no scratch: a workgroup stages the dx of one or two chains in LDS and walks the flat entry stream (k_sparse.hip: k_sparse_quad);
// T != null: tree states (X = heights, Rt = rates, ld = their row stride; logjac may be null), else X = plain vectors; ll and qout
// (the quadratic form dx^T P dx itself; what the Metropolis-Hastings driver keeps per chain) may each be null
bool sparse_quad_available(const SparseDev& S, int64_t batch);
hipError_t launch_sparse_quad(const SparseDev& S, const SparseTreeDev* T, const double* X, const double* Rt, int64_t ld, const double* tH,
                              const double* rMu, int64_t batch, double* ll, double* logjac, double* qout, hipStream_t st);

// Metropolis-coupled MCMC (k_mc3.hip): the temperature rank of every GLOBAL chain, the ladder of reciprocal temperatures and the
// swap counters per rung; all pointers are device memory.
struct Mc3Dev {
    int n_chains;                           // chains per group (NChains)
    int64_t total;                          // global number of chains (a multiple of n_chains)
    const double* ladder;                   // [n_chains] reciprocal temperatures, ladder[0] = 1
    int32_t* rank;                          // [total] temperature rank of every global chain
    unsigned long long *tried, *accepted;   // [n_chains - 1] swaps between the ranks i and i + 1
};

// Incremental likelihood of the two-launch Metropolis-Hastings path on large trees (k_mh_inc.hip): distances and z = L^-1 (d - mu) of
// every chain's CURRENT state, z' of the pending proposal; mode = how the pending proposal's z' reaches zcur when it is accepted.
struct MhInc {
    double* X0;          // [batch][ldx] distances of the current states
    double* zcur;        // [batch][NPz]
    double* zprop;       // [batch][NPz] z' of a pending sparse proposal
    const double* zt;    // tile-major z' of a pending dense proposal (k_split's scratch), nr rows per tile
    int NPz, nr;
    int mode;            // of the PENDING proposal: 0 likelihood not moved, 1 sparse (zprop), 2 dense (zt)
    int prop_mode;       // of the proposal k_mh_step_wg is about to make: 0 / 1: it writes ll' itself (1: by columns of L^-1), 2: the row-split launch follows
};
// a run of consecutive steps without a dense proposal, every chain's state in LDS from the first to the last (k_mh_segment.hip)
struct MhSegPending {          // a dense proposal that is still to be decided when the segment starts (k_mh_step_wg proposed it into H1 / R1 /
    int p_acc;                 // sc1 / X1 / post1 / pcomp1 / lnqj, the row-split launch left its ln likelihood and its z tiles): -1 = none
    int jac_root;
    int accumulate;            // that step closes an iteration: the node ages go into the running sums
    uint64_t step;             // its step number (the acceptance uniform's stream)
    double* trace_alpha;       // [batch] or null
    int8_t* trace_accept;
    const double* X1;          // [batch][n] its distances
    int z_in_zprop;            // its z' is in MhInc::zprop (batches beyond 1024 chains: taken there chunk by chunk), not in the z tiles
    // ... and the dense proposal that FOLLOWS the segment's last step: proposed by the segment's launch from the state it holds in LDS (what a
    // launch of k_mh_step_wg would do after reading everything back: H1 / R1 / sc1 / post1 / pcomp1 / lnqj / the summands / its distances)
    int p_tail;                // its row of the proposal table, -1 = none
    double* X1_tail;           // [batch][n] its distances
    int ahead_from;            // trees from this many nodes: the chain wave draws the next step's proposal while the step in flight is evaluated
    int prior_draws;           // the clock prior wave draws the next step's proposal (segment kernels with prior waves; mh_segment_device.hpp: SegSpec)
};
bool mh_segment_available(const MhDev& M, const MvnDev& V);
// the same over a sparse precision matrix (k_mh_segment_sparse.hip); I: X0 = current distances [batch][n], zcur / zprop = the quadratic
// forms q [batch] of the current states / of the pending dense proposal (NPz = 1)
bool mh_segment_sparse_available(const MhDev& M, const SparseDev& Sp);
int mh_segment_sparse_list();      // moved distances of one proposal at most
hipError_t launch_mh_segment_sparse(const MhDev& M, const SparseDev& Sp, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched,
                                    int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept,
                                    int64_t gs_base, int summands_kept, const MhSegPending* pending, int list_all, hipStream_t st);
hipError_t launch_mh_segment(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const MhInc& I, const int32_t* sched,
                             int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept,
                             int64_t gs_base, int summands_kept, const MhSegPending* pending, hipStream_t st);
hipError_t launch_mh_inc_init(const MhDev& M, const TreeDev& T, const MhInc& I, int n_dim, int64_t ldx, hipStream_t st);   // X0 from the current states
hipError_t launch_mh_inc_take_z(const MhInc& I, double* dst, int64_t b0, int64_t count, hipStream_t st);                 // dst[b0 ..] <- zt of `count` chains

// Workspace of the device leapfrog (k_hmc.hip); all pointers are device memory.
struct HmcDev {
    int n_nodes, dim, root_right;
    int64_t batch, ld;
    const int32_t *pos_field, *pos_index;   // [dim] position layout (toVector order)
    double *sc, *H, *R;                     // state: scalars [5][batch] (birth, death, tH, rMu, rVar), heights / rates [batch][ld]
    double *lp, *gp_sc, *gp_H, *gp_R;       // prior: value [batch], gradient wrt the scalars [5][batch], heights, rates
    double *ll, *gl_H, *gl_R, *gl_tH, *gl_rMu;   // likelihood: value and gradient
    double *q, *p, *grad, *value;           // [batch][dim] position, momentum, gradient; [batch] ln target
    const double *eps, *dir, *inv_mass;     // [batch] step sizes, [batch] +-1 or null, [dim] inverse masses
};

// Per-chain state of the device NUTS (k_nuts.hip); all pointers are device memory, position layout [batch][dim] unless noted.
struct NutsDev {
    int max_depth;                          // levels of the first-leaf stack
    double *qm, *pm, *gm, *qp, *pp, *gp;    // the tree's two ends: position, momentum, gradient
    double *qc, *gc, *lpc;                  // candidate inside the sub tree being built (+ ln target [batch])
    double *qn, *gn, *lpn;                  // the transition's current proposal
    double *sq, *sp;                        // [batch][max_depth][dim] first leaf (position, momentum) of the open sub tree of 2^k leaves
    double *log_u, *joint0, *alpha;         // [batch] slice variable, -H at the start, sum of min(1, exp(H0 - H)) over the leaves
    int *j, *v, *i, *n, *n1, *s1, *done, *n_alpha, *depth, *leaf;   // [batch] doubling, direction, leaf index, counts, flags
};
hipError_t launch_nuts_begin(const HmcDev& D, const NutsDev& N, uint64_t seed, int64_t chain0, uint64_t transition, hipStream_t st);
hipError_t launch_nuts_step(const HmcDev& D, const NutsDev& N, uint64_t seed, int64_t chain0, uint64_t transition, int max_depth, int* active,
                            hipStream_t st);
hipError_t launch_nuts_end(const HmcDev& D, const NutsDev& N, hipStream_t st);

int padded_blocks(int n);          // supported R for dimension n, or -1
int sweep_chunk_columns(int R);    // columns per register buffer (ncols granularity)

// Many chains per launch: multiply form on the fp64 matrix cores (k_wide.hip).  launch_logpdf / launch_tree_logpdf
// route to these when use_wide() says so; MCD_WIDE=0 / 1 forces the choice, MCD_WIDE_CT=1|2|4 the chains per workgroup / 16.
hipError_t launch_logpdf_wide(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st);
hipError_t launch_tree_logpdf_wide(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                   const double* rMu, int64_t batch, double* ll, double* logjac, hipStream_t st);
hipError_t launch_grad_wide(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                            hipStream_t st);
hipError_t launch_tree_grad_wide(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                 const double* rMu, int64_t batch, double* ll, double* gH, double* gR, double* gtH, double* grMu,
                                 hipStream_t st);
hipError_t launch_grad_wide_mc(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg, hipStream_t st);
hipError_t launch_tree_grad_wide_mc(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                    const double* rMu, int64_t batch, double* ll, double* gH, double* gR, double* gtH, double* grMu,
                                    hipStream_t st);
// Row blocks of W split over G workgroups per chain tile (k_split.hip): N > 128 up to 1024 chains, raw x and tree states.
// The cross-workgroup scratch (G x 16 partial sums and a counter per tile) is taken from the handle's pool (M.split), one set
// per stream, or per (capture, stream) while a stream is being captured.
constexpr int64_t kSplitMaxBatch = 1024;
constexpr int kSplitMaxG = 32;
constexpr size_t kSplitScratchDoubles = (kSplitMaxBatch / 16) * kSplitMaxG * 16;
constexpr size_t kSplitCounters = kSplitMaxBatch / 16;
SplitHost* split_host_create(int n, const double* W_rowmajor, hipError_t* err);   // W = L^-1; tile streams on the current device + scratch pool
void split_host_destroy(SplitHost* s);
hipError_t split_release_stream(SplitHost* s, hipStream_t st);   // the stream's eager scratch set back to the pool (mcd_mvn_release_stream)
bool use_split(const MvnDev& M, int64_t batch);
hipError_t launch_logpdf_split(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st);
hipError_t launch_tree_logpdf_split(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                    const double* rMu, int64_t batch, double* ll, double* logjac, hipStream_t st);
hipError_t launch_grad_split(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg, hipStream_t st);
hipError_t launch_logpdf_split_z(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, const double** zt, int* nr, hipStream_t st);
hipError_t launch_tree_grad_split(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                  const double* rMu, int64_t batch, double* ll, double* gH, double* gR, double* gtH, double* grMu, hipStream_t st);
bool use_split_grad(const MvnDev& M, int64_t batch);
int effective_form(const MvnDev& M);   // MCD_FORM_* in force for this handle
hipError_t prepare_wide();            // per-device attribute set-up of the multiply-form kernels (current device)
hipError_t prepare_wide_grad();
hipError_t prepare_wide_grad_mc();
bool use_wide_grad(const MvnDev& M, int64_t batch);
int wide_chain_tiles(int64_t batch);
bool use_wide(const MvnDev& M, int64_t batch);
int set_logpdf_form(int form);
hipError_t launch_logpdf(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, hipStream_t st);
hipError_t launch_grad(const MvnDev& M, const double* X, int64_t ldx, int64_t batch, double* ll, double* G, int64_t ldg,
                       hipStream_t st);
hipError_t launch_tree_logpdf(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                              const double* tH, const double* rMu, int64_t batch, double* ll, double* logjac,
                              hipStream_t st);
hipError_t launch_tree_grad(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds,
                            const double* tH, const double* rMu, int64_t batch, double* ll, double* gH, double* gR,
                            double* gtH, double* grMu, hipStream_t st);

hipError_t launch_prior(const PriorDev& P, const double* birth, const double* death, const double* tH, const double* H,
                        const double* rMu, const double* rVar, const double* Rt, int64_t lds, int64_t batch, double* lp,
                        double* comp, hipStream_t st);

hipError_t launch_prior_grad(const PriorDev& P, const double* birth, const double* death, const double* tH, const double* H,
                             const double* rMu, const double* rVar, const double* Rt, int64_t lds, int64_t batch, double* lp,
                             double* g_birth, double* g_death, double* g_tH, double* g_H, double* g_rMu, double* g_rVar, double* g_R,
                             hipStream_t st);
hipError_t launch_hmc_kick(const HmcDev& D, double kick, int use_pos_grad, hipStream_t st);   // p += kick eps grad(q)
hipError_t launch_hmc_scatter(const HmcDev& D, hipStream_t st);               // state arrays <- D.q
hipError_t launch_hmc_drift(const HmcDev& D, hipStream_t st);                 // q += eps Minv p, scattered into the state
hipError_t launch_hmc_collect(const HmcDev& D, hipStream_t st);               // q, grad, value from the state and the gradient kernels
hipError_t launch_hmc_kick_drift(const HmcDev& D, double kick, hipStream_t st);     // kick + drift in one launch (gradient from the kernels' outputs)
hipError_t launch_hmc_kick_collect(const HmcDev& D, double kick, hipStream_t st);   // closing kick + collect
// one row of the proposal table on the host (k_mh.hip turns it into a kernel argument)
struct MhRow {
    int kind, node, n1, n2, jac_root;
    double p0, p1;
};
// accept the pending step of proposal p_acc (< 0: none) and propose proposal p_prop (< 0: none) with the ln prior of its proposed state
hipError_t launch_mh_step(const MhDev& M, const PriorDev& P, int p_acc, int jac_root_acc, int p_prop, const MhRow& row_prop, int draw_slot,
                          uint64_t step_acc, uint64_t seed, int accumulate_now, double* trace_alpha, int8_t* trace_accept, int prior_inline,
                          const TreeDev* T, int n_dim, double* X1, int64_t ldx, hipStream_t st, const MhInc* inc = nullptr, const MvnDev* V = nullptr,
                          int summands_init = -1);   // 1: MhDev::psum does not hold the current states' summands yet (-1: when nothing is pending)
// true: launch_mh_step takes the workgroup-per-chain kernel, which can also leave the proposed states' distances in X1 [batch][ldx]
// (T, n_dim, X1 given) for a plain-vector likelihood launch
bool mh_step_wg_active(const MhDev& M, int prior_inline, int min_nodes = 320);   // (default: trees of more than min_nodes nodes)
hipError_t launch_mh_tune(const MhDev& M, hipStream_t st);
hipError_t launch_mc3_swap(const Mc3Dev& C, const double* lnpost, int world, int64_t per_rank, int n_swaps, uint64_t seed, uint64_t phase,
                           double* beta_local, int64_t chain0, int64_t batch, hipStream_t st);
// ln prior of the proposed states from pflags / pcomp (what launch_mh_step leaves when asked not to evaluate it itself) as extra
// workgroups of the sweep's tree-likelihood launch (k_tree_logpdf.hip): the ln prior and the ln likelihood of a proposal depend
// on nothing but the proposal
bool tree_logpdf_can_carry_prior(const MvnDev& M, int64_t batch, int n_nodes);
hipError_t launch_tree_logpdf_with_prior(const MvnDev& M, const TreeDev& T, const double* H, const double* Rt, int64_t lds, const double* tH,
                                         const double* rMu, int64_t batch, double* ll, double* logjac, const MhDev& J, const PriorDev& JP,
                                         hipStream_t st);
// state-independent draws of the steps [idx0, idx0 + count) of the schedule (count <= 64), one thread per (step, chain)
hipError_t launch_mh_draws(const MhDev& M, const int32_t* sched, int64_t idx0, int count, uint64_t step0, uint64_t seed, hipStream_t st);
// whole schedule in one launch for trees of 65 .. 320 nodes at up to 1024 chains (k_mh_chain_big.hip): two chains per workgroup,
// the factor streamed through the sweep's LDS ring once per step
bool mh_chain_big_available(const MhDev& M, const MvnDev& V);
hipError_t launch_mh_chain_big(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const int32_t* sched, int64_t n_steps,
                               int32_t S, int accumulate, uint64_t step0, uint64_t seed, double* trace_alpha, int8_t* trace_accept,
                               hipStream_t st);
// whole schedule in one launch (k_mh_chain.hip); needs n_nodes <= 64 and mh_chain_lds_bytes(...) <= 64 KB
size_t mh_chain_lds_bytes(int n, int n_prop, int wpb);
hipError_t launch_mh_chain(const MhDev& M, const MvnDev& V, const TreeDev& T, const PriorDev& P, const double* Fp,
                           const int32_t* sched, int64_t n_steps, int32_t S, int accumulate, uint64_t step0, uint64_t seed,
                           double* trace_alpha, int8_t* trace_accept, hipStream_t st);

}  // namespace mcd

// handle internals shared between the translation units of the C ABI (mvn_capi.cpp, prior_capi.cpp, mh_capi.cpp)
struct mcd_tree;
struct mcd_prior;
int mcd_tree_internal_(const mcd_tree* t, const mcd::MvnDev** mvn, const mcd::TreeDev** tree, int* device, const int32_t** host_parent,
                       const double** host_L);
int mcd_prior_internal_(const mcd_prior* p, const mcd::PriorDev** prior, int* device);
