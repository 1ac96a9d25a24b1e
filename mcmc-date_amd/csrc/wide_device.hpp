// wide_device.hpp -- shared device code of the multiply-form kernels (k_wide.hip, k_wide_grad.hip): tile geometry,
// the staging of the residuals R = x - mu (or tree distances - mu) into LDS, the streamed triangular pass.
#pragma once
#include "mvn_kernels.h"
#include <type_traits>

namespace mcd {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int WD_WAVES = 8;
constexpr int WD_SB = 256;          // rows per super block = columns per staged chunk
constexpr int WD_LD = WD_SB + 2;    // LDS row stride in doubles: 516 dwords = 4 (mod 64 banks)
constexpr int WD_P = 8;             // W tiles in flight per row block

// diagnostic build (make stamp_wide): s_memtime phase stamps of every wave of workgroup 0, read back with
// mcd_wide_debug_stamps (tools/microbench/wide_stamps.py)
#ifdef MCD_WIDE_STAMP
__device__ unsigned long long g_wide_dbg[WD_WAVES * 16];
#define WD_T(i) do { if (blockIdx.x == 0 && lane == 0) g_wide_dbg[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WD_T(i) do { } while (0)
#endif

// Sum over the 64 lanes with DPP row operations + four readlanes, result wave-uniform, fixed order (as wave_sum in
// mvn_device.hpp): about a tenth of the cost of six ds_bpermute exchanges.
template <int CTRL>
__device__ __forceinline__ double wd_dpp_mov64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wd_readlane64(double v, int srclane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wd_wave_sum(double v)
{
    v += wd_dpp_mov64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += wd_dpp_mov64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += wd_dpp_mov64<0x124>(v);  // row_ror:4
    v += wd_dpp_mov64<0x128>(v);  // row_ror:8 -> every lane holds its 16-lane row sum
    return (wd_readlane64(v, 0) + wd_readlane64(v, 16)) + (wd_readlane64(v, 32) + wd_readlane64(v, 48));
}

struct WideSrc {
    const double* X;                // raw x: [batch][ldx]                                       (TREE = false)
    int64_t ldx;
    TreeDev T;                      // tree state: heights / rates [batch][lds], tH, rMu [batch]   (TREE = true)
    const double *H, *Rt;
    int64_t lds;
    const double *tH, *rMu;
    double* logjac;
};

// One 256-column chunk of R into LDS.  512 threads = two chain rows per pass, so a thread keeps its column: the
// per-column operands (mu, the slot's node and its parent) are read once, and the CT * 8 passes over the chains are
// unrolled so that all their loads are in flight together.
template <int CT, bool TREE>
__device__ __forceinline__ void wide_stage(double* rs, const double* scs, const MvnDev& M, const WideSrc& A, int64_t b0, int64_t batch,
                                           int kc0, int tid)
{
    const int j = tid & (WD_SB - 1), k = kc0 + j, ch0 = tid >> 8;
    const bool live = k < M.n;
    const double m = live ? M.mu[k] : 0.0;
    if constexpr (!TREE) {
        constexpr int G = 8;                                   // chain rows in flight per thread
#pragma unroll 1
        for (int g = 0; g < CT * 8; g += G) {
            double v[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int64_t b = b0 + ch0 + 2 * (g + i);
                v[i] = (live && b < batch) ? A.X[b * A.ldx + k] : m;        // padded columns and chains: exact zeros
            }
#pragma unroll
            for (int i = 0; i < G; ++i) rs[(ch0 + 2 * (g + i)) * WD_LD + j] = v[i] - m;   // dxs = xs - mu  (app/Probability.hs:171)
        }
    } else {
        // distances from the tree state -- app/Probability.hs:201-207 (as load_tree in mvn_device.hpp); scs[] holds
        // tH * rMu of the workgroup's chains.  Slot 0 (the two root branches, sumFirstTwo) is left to wide_stage_root.
        constexpr int G = CT == 4 ? 16 : 8;
        const int a = live ? A.T.slot_node[k] : 0;
        const int pa = live ? A.T.slot_parent[k] : 0;
#pragma unroll 1
        for (int g = 0; g < CT * 8; g += G) {
            double hp[G], ha[G], ra[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int64_t b = (b0 + ch0 + 2 * (g + i) < batch) ? b0 + ch0 + 2 * (g + i) : batch - 1;
                const double* h = A.H + b * A.lds;
                hp[i] = h[pa];
                ha[i] = h[a];
                ra[i] = A.Rt[b * A.lds + a];
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int ch = ch0 + 2 * (g + i);
                const double d = ((hp[i] - ha[i]) * ra[i]) * scs[ch];
                if (k != 0) rs[ch * WD_LD + j] = (live && b0 + ch < batch) ? d - m : 0.0;
            }
        }
    }
}

// The same for one chunk of a tree state, keeping each slot's branch duration t = h_parent - h_node and rate per chain
// row in registers: the gradient's chain rule (k_wide_grad.hip) needs them again and would otherwise gather them twice.
template <int CT>
__device__ __forceinline__ void wide_stage_tree_keep(double* rs, const double* scs, const MvnDev& M, const WideSrc& A, int64_t b0,
                                                     int64_t batch, int tid, double (&tk)[CT * 8], double (&rk)[CT * 8])
{
    const int j = tid & (WD_SB - 1), ch0 = tid >> 8;
    const bool live = j < M.n;
    const double m = live ? M.mu[j] : 0.0;
    const int a = live ? A.T.slot_node[j] : 0;
    const int pa = live ? A.T.slot_parent[j] : 0;
    auto rows8 = [&](auto half) {
        constexpr int h0 = decltype(half)::value * 8;
        double hp[8], ha[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t b = (b0 + ch0 + 2 * (h0 + i) < batch) ? b0 + ch0 + 2 * (h0 + i) : batch - 1;
            const double* h = A.H + b * A.lds;
            hp[i] = h[pa];
            ha[i] = h[a];
            rk[h0 + i] = A.Rt[b * A.lds + a];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = ch0 + 2 * (h0 + i);
            const bool in = live && b0 + ch < batch;
            tk[h0 + i] = in ? hp[i] - ha[i] : 0.0;
            rk[h0 + i] = in ? rk[h0 + i] : 0.0;
            const double d = (tk[h0 + i] * rk[h0 + i]) * scs[ch];
            if (j != 0) rs[ch * WD_LD + j] = in ? d - m : 0.0;
        }
    };
    rows8(std::integral_constant<int, 0>{});
    if constexpr (CT == 2) {
        __builtin_amdgcn_sched_barrier(0);
        rows8(std::integral_constant<int, 1>{});
    }
}

// slot 0 of the distances: the branches of the two children of the root added up (app/Tools.hs:36-48), one chain per
// thread; also the root-branch Jacobian.
template <int CT>
__device__ __forceinline__ void wide_stage_root(double* rs, const double* scs, const MvnDev& M, const WideSrc& A, int64_t b0,
                                                int64_t batch, bool first, int tid)
{
    if (tid >= CT * 16) return;
    const int64_t b = (b0 + tid < batch) ? b0 + tid : batch - 1;
    const double* h = A.H + b * A.lds;
    const double* r = A.Rt + b * A.lds;
    const int a = A.T.slot_node[0], pa = A.T.slot_parent[0], rr = A.T.root_right;
    double d = (h[pa] - h[a]) * r[a];
    d = d + (h[0] - h[rr]) * r[rr];
    d = d * scs[tid];
    const bool in = b0 + tid < batch;
    if (first && in && A.logjac != nullptr) A.logjac[b] = log(1.0 / d);      // app/Probability.hs:394, 409
    rs[tid * WD_LD] = in ? d - M.mu[0] : 0.0;
}

// One row block against a run of k tiles: acc[ct] += W-tile(j) x R-tile(kt0 + j), j = 0 .. nt - 1 (nt a multiple of 4,
// wave-uniform).  `w` points at this lane's element of the first tile; 8 tiles are kept in flight (unconditional loads,
// index clamped to the last tile).
template <int CT>
__device__ __forceinline__ void wide_tri_pass(const double* __restrict__ w, int nt, int kt0, const double* rs, int col, int kq,
                                              d4 (&acc)[CT])
{
    if (nt <= 0) return;
    const int last = nt - 1;
    double ring[WD_P];
#pragma unroll
    for (int p = 0; p < WD_P; ++p) ring[p] = w[(p < last ? p : last) * 64];
    int j0 = 0;
    for (; j0 + WD_P <= nt; j0 += WD_P) {
#pragma unroll
        for (int p = 0; p < WD_P; ++p) {
            const int j = j0 + p, nx = j + WD_P;
            const double a = ring[p];
            ring[p] = w[(nx < last ? nx : last) * 64];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rs[(ct * 16 + col) * WD_LD + (kt0 + j) * 4 + kq], acc[ct], 0, 0, 0);
        }
    }
    if (j0 < nt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[p], rs[(ct * 16 + col) * WD_LD + (kt0 + j0 + p) * 4 + kq], acc[ct], 0, 0, 0);
        }
    }
}

}  // namespace mcd
