// host_factor.cpp -- one-time host preparation of the likelihood operands.
//
// The reference prepares (mu, Sigma^-1, logdet Sigma) once in `prepare` (app/Main.hs:207-208,
// 230-231) and closes over them (app/Main.hs:333-347).  Here the same one-time step produces
// the Cholesky factor of Sigma and the two scaled, packed triangular factors the kernels stream.
// Inner products are accumulated in long double: this runs once per analysis, accuracy wins.
#include "host_factor.h"
#include "split_sched.hpp"

#include <cmath>
#include <cstring>

namespace mcd {

bool cholesky_lower(int n, const std::vector<double>& A, std::vector<double>& L)
{
    L.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) {
        double* Li = &L[(size_t)i * n];
        for (int j = 0; j <= i; ++j) {
            const double* Lj = &L[(size_t)j * n];
            long double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= (long double)Li[k] * (long double)Lj[k];
            if (i == j) {
                if (!(s > 0.0L) || !std::isfinite((double)s)) return false;
                Li[i] = (double)sqrtl(s);
            } else {
                Li[j] = (double)(s / (long double)Lj[j]);
            }
        }
    }
    return true;
}

// inverse of a lower-triangular matrix (row-major), result lower-triangular
static void tri_lower_inverse(int n, const std::vector<double>& C, std::vector<double>& W)
{
    W.assign((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        W[(size_t)j * n + j] = 1.0 / C[(size_t)j * n + j];
        for (int i = j + 1; i < n; ++i) {
            long double s = 0.0L;
            const double* Ci = &C[(size_t)i * n];
            for (int k = j; k < i; ++k) s -= (long double)Ci[k] * (long double)W[(size_t)k * n + j];
            W[(size_t)i * n + j] = (double)(s / (long double)Ci[i]);
        }
    }
}

bool precision_factors(int n, const std::vector<double>& P, std::vector<double>& W, std::vector<double>& L)
{
    // P = W^T W with W lower triangular: the Cholesky factor of P with rows and columns reversed, reversed back and
    // transposed (J P J = C C^T, W = J C^T J).  No matrix is inverted on the way to W -- the factor the multiply-form kernels
    // stream -- so its error is that of one Cholesky factorisation; L = W^-1 (Sigma = L L^T), which the column sweeps
    // stream, costs one triangular inversion: errors grow with cond(W) = sqrt(cond(P)), not with cond(P)^2 as through
    // Sigma = P^-1 followed by a second factorisation.
    std::vector<double> R((size_t)n * n), C;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) R[(size_t)i * n + j] = P[(size_t)(n - 1 - i) * n + (n - 1 - j)];
    if (!cholesky_lower(n, R, C)) return false;
    W.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) W[(size_t)i * n + j] = C[(size_t)(n - 1 - j) * n + (n - 1 - i)];
    tri_lower_inverse(n, W, L);
    return true;
}

bool spd_inverse(int n, const std::vector<double>& P, std::vector<double>& S)
{
    std::vector<double> C, W;
    if (!cholesky_lower(n, P, C)) return false;      // P = C C^T
    tri_lower_inverse(n, C, W);                      // W = C^-1
    S.assign((size_t)n * n, 0.0);                    // S = W^T W
    // column-of-W dot products: transpose W first so the inner loop is contiguous
    std::vector<double> Wt((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Wt[(size_t)j * n + i] = W[(size_t)i * n + j];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            long double s = 0.0L;
            const double* a = &Wt[(size_t)i * n];
            const double* b = &Wt[(size_t)j * n];
            for (int k = i; k < n; ++k) s += (long double)a[k] * (long double)b[k];  // W[k][i] = 0 for k < i
            S[(size_t)i * n + j] = S[(size_t)j * n + i] = (double)s;
        }
    }
    return true;
}

size_t packed_index(int R, int row, int col)
{
    const int k = row >> 6, lane = row & 63;
    return ((((size_t)(col >> 1) * R + k) * 64 + lane) << 1) + (size_t)(col & 1);
}

void pack_factors(int n, int R, const std::vector<double>& L, std::vector<double>& mu_pad, const double* mu,
                  std::vector<double>& invdiag, std::vector<double>& Ft, std::vector<double>& Ut)
{
    const int NP = 64 * R;
    mu_pad.assign(NP, 0.0);
    invdiag.assign(NP, 1.0);
    Ft.assign((size_t)NP * NP, 0.0);
    Ut.assign((size_t)NP * NP, 0.0);
    for (int i = 0; i < n; ++i) {
        mu_pad[i] = mu[i];
        invdiag[i] = 1.0 / L[(size_t)i * n + i];
    }
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < i; ++j) {
            const double lij = L[(size_t)i * n + j];
            // forward:  row i, column j  ->  L_ij / L_ii
            Ft[packed_index(R, i, j)] = lij * invdiag[i];
            // backward: U = L^T; row j, column i  ->  U_ji / U_jj = L_ij / L_jj
            Ut[packed_index(R, j, i)] = lij * invdiag[j];
        }
    }
}

void invert_factor(int n, const std::vector<double>& L, std::vector<double>& W)
{
    // rows of W one after the other: W_i: = (e_i - sum_{k<i} L_ik W_k:) / L_ii, long double accumulation
    W.assign((size_t)n * n, 0.0);
    std::vector<long double> acc(n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < i; ++j) acc[j] = 0.0L;
        const double* Li = &L[(size_t)i * n];
        for (int k = 0; k < i; ++k) {
            const long double lik = Li[k];
            const double* Wk = &W[(size_t)k * n];
            for (int j = 0; j <= k; ++j) acc[j] -= lik * (long double)Wk[j];
        }
        const long double d = Li[i];
        for (int j = 0; j < i; ++j) W[(size_t)i * n + j] = (double)(acc[j] / d);
        W[(size_t)i * n + i] = (double)(1.0L / d);
    }
}

// one A-operand tile of v_mfma_f64_16x16x4_f64: row block ib, k tile kt; lane l = W[16 ib + (l & 15)][4 kt + (l >> 4)]
static void w_tile(int n, const std::vector<double>& W, int ib, int kt, double* t)
{
    for (int l = 0; l < 64; ++l) {
        const int row = 16 * ib + (l & 15), colk = 4 * kt + (l >> 4);
        t[l] = (row < n && colk <= row) ? W[(size_t)row * n + colk] : 0.0;
    }
}

void build_split_schedule(int n, int G, const std::vector<double>& W, SplitScheduleHost& out)
{
    const int NB = (n + 15) / 16;
    out.G = G;
    out.NB = NB;
    out.nc = (16 * NB + 255) / 256;
    out.base.assign(G, 0);
    out.Ws.clear();
    out.Ws.reserve(((size_t)2 * NB * (NB + 1) + 2) * 64);
    int toff = 0;
    for (int g = 0; g < G; ++g) {
        out.base[g] = toff;
        // the group's stream: block after block (split_sched.hpp), k ascending; the four runs are consecutive pieces of it,
        // so packing the stream in order packs every wave's run in the order the wave consumes it
        for (int r = 0; r * G < NB; ++r) {
            const int b = sp_block(G, g, r);
            if (b >= NB) continue;
            for (int kt = 0; kt < 4 * (b + 1); ++kt, ++toff) {
                out.Ws.resize(out.Ws.size() + 64);
                w_tile(n, W, b, kt, &out.Ws[out.Ws.size() - 64]);
            }
        }
    }
    // tiles travel in pairs: lane l's element of tile 2 i, then of tile 2 i + 1 (one 16-byte load per lane); every wave's run
    // is a whole number of pairs (multiples of 4 tiles)
    for (size_t t = 0; t + 1 < (size_t)toff; t += 2) {
        double tmp[128];
        double* p = &out.Ws[t * 64];
        for (int l = 0; l < 64; ++l) {
            tmp[2 * l] = p[l];
            tmp[2 * l + 1] = p[64 + l];
        }
        std::memcpy(p, tmp, sizeof tmp);
    }
    out.Ws.resize(out.Ws.size() + 128, 0.0);                       // one zero pair: the clamped prefetch of an empty run reads it
}

void pack_w_tiles(int n, const std::vector<double>& W, std::vector<double>& Wt, std::vector<double>& Wtb)
{
    const int NB = (n + 15) / 16;
    Wt.assign((size_t)2 * NB * (NB + 1) * 64, 0.0);
    for (int ib = 0; ib < NB; ++ib)
        for (int kt = 0; kt < 4 * (ib + 1); ++kt) {
            double* t = &Wt[((size_t)2 * ib * (ib + 1) + kt) * 64];
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * ib + (l & 15), colk = 4 * kt + (l >> 4);
                if (row < n && colk <= row) t[l] = W[(size_t)row * n + colk];
            }
        }
    Wtb.assign((size_t)2 * NB * (NB + 1) * 64, 0.0);      // sum over ib of 4 (NB - ib) tiles = 2 NB (NB + 1)
    for (int ib = 0; ib < NB; ++ib)
        for (int kt = 4 * ib; kt < 4 * NB; ++kt) {
            double* t = &Wtb[((size_t)4 * (ib * NB - ib * (ib - 1) / 2) + (kt - 4 * ib)) * 64];
            for (int l = 0; l < 64; ++l) {
                const int row = 4 * kt + (l >> 4), colk = 16 * ib + (l & 15);     // element (row, colk) of W
                if (row < n && colk <= row) t[l] = W[(size_t)row * n + colk];
            }
        }
}

}  // namespace mcd

// Host-only self test of the row-split schedule (tests/test_host.py, no GPU): runs k_split.hip's walk -- the arithmetic of
// split_sched.hpp over the tile stream build_split_schedule packed: runs, whole and cut blocks, LDS slots, the fixed-order
// combination -- on the CPU for one random residual vector and returns the relative difference between the sum of squares it
// finds and |W r|^2 computed directly; < 0 for a structural fault.
extern "C" double mcd_split_schedule_selftest_(int n, int G, unsigned seed)
{
    if (n < 1 || G < 1 || G > 32) return -1.0;
    std::vector<double> W((size_t)n * n, 0.0), r(((size_t)n + 15) / 16 * 16 + 16, 0.0);
    unsigned long long st = 0x9E3779B97F4A7C15ull ^ seed;
    auto rnd = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        return (double)((st >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53) - 0.5;
    };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) W[(size_t)i * n + j] = rnd();
    for (int i = 0; i < n; ++i) r[i] = rnd();
    mcd::SplitScheduleHost h;
    mcd::build_split_schedule(n, G, W, h);
    const int NB = (n + 15) / 16;
    if (h.nc != (16 * NB + 255) / 256 || h.NB != NB) return -2.0;
    long tiles_seen = 0;
    long double total = 0.0L;
    int fault = 0;
    for (int g = 0; g < G; ++g) {
        const mcd::SpGroup q = mcd::sp_group(NB, G, g);
        if ((h.base[g] & 3) || mcd::sp_run_start(q, mcd::SP_NW) != q.Tg) return -3.0;
        std::vector<std::vector<double>> slot(mcd::SP_NSLOT, std::vector<double>(16, 0.0));
        std::vector<bool> slot_set(mcd::SP_NSLOT, false), used(mcd::SP_NSLOT, false);
        long double ss = 0.0L;
        for (int w = 0; w < mcd::SP_NW; ++w) {
            int pos = mcd::sp_run_start(q, w);                                     // stream position of the run's next tile
            int nseg = 0;
            mcd::sp_for_each_segment(NB, G, g, q, w, [&](int k0, int nt, int kind) {
                if (nt <= 0 || (nt & 3) || (k0 & 3) || kind < 0 || kind > 2) fault = 4;
                if (4 * (k0 + nt) > q.ncols) fault = 5;                            // reads a column the group does not stage
                if (++nseg > mcd::SPH_MAXSEG) fault = 3;
                double acc[16] = {0};
                for (int j = 0; j < nt; ++j, ++pos) {
                    const size_t t = (size_t)h.base[g] + pos;
                    const double* pair = &h.Ws[(t >> 1) * 128];
                    for (int l = 0; l < 64; ++l) acc[l & 15] += pair[2 * l + (t & 1)] * r[4 * (k0 + j) + (l >> 4)];
                }
                tiles_seen += nt;
                if (kind == 0) {
                    for (int i = 0; i < 16; ++i) ss += (long double)acc[i] * acc[i];
                } else {
                    const int sl = kind == 1 ? 2 * w - 1 : 2 * w;
                    if (sl < 0 || sl >= mcd::SP_NSLOT || slot_set[sl]) {
                        fault = 6;
                        return;
                    }
                    slot_set[sl] = true;
                    for (int i = 0; i < 16; ++i) slot[sl][i] = acc[i];
                }
            });
            if (pos != mcd::sp_run_start(q, w + 1)) fault = 7;
        }
        mcd::sp_for_each_cut(NB, G, g, q, [&](int wf, int wl, int blk) {
            if (wf < 0 || wl >= mcd::SP_NW || wl <= wf || blk < 0 || blk >= NB) {
                fault = 9;
                return;
            }
            double z[16] = {0};
            for (int w = wf; w <= wl; ++w) {
                const int sl = w == wf ? 2 * wf : 2 * w - 1;
                if (!slot_set[sl] || used[sl]) {
                    fault = 10;
                    continue;
                }
                used[sl] = true;
                for (int k = 0; k < 16; ++k) z[k] += slot[sl][k];
            }
            for (int k = 0; k < 16; ++k) ss += (long double)z[k] * z[k];
        });
        for (int sl = 0; sl < mcd::SP_NSLOT; ++sl)
            if (slot_set[sl] && !used[sl]) fault = 11;
        total += ss;
    }
    if (fault) return -(double)fault;
    if (tiles_seen != 2L * NB * (NB + 1)) return -12.0;
    long double direct = 0.0L;
    for (int i = 0; i < n; ++i) {
        long double z = 0.0L;
        for (int j = 0; j <= i; ++j) z += (long double)W[(size_t)i * n + j] * r[j];
        direct += z * z;
    }
    return (double)(fabsl(total - direct) / direct);
}
