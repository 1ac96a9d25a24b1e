// host_factor.cpp -- one-time host preparation of the likelihood operands.
//
// The reference prepares (mu, Sigma^-1, logdet Sigma) once in `prepare` (app/Main.hs:207-208,
// 230-231) and closes over them (app/Main.hs:333-347).  Here the same one-time step produces
// the Cholesky factor of Sigma and the two scaled, packed triangular factors the kernels stream.
// Inner products are accumulated in long double: this runs once per analysis, accuracy wins.
#include "host_factor.h"

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

void pack_w_tiles(int n, const std::vector<double>& L, std::vector<double>& Wt, std::vector<double>& Wtb)
{
    // rows of W one after the other: W_i: = (e_i - sum_{k<i} L_ik W_k:) / L_ii, long double accumulation
    std::vector<double> W((size_t)n * n, 0.0);
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
