// host_factor.h -- one-time host preparation of the likelihood operands (internal).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mcd {

// Lower Cholesky factor of a symmetric matrix (row-major n x n).  false if not positive definite.
bool cholesky_lower(int n, const std::vector<double>& A, std::vector<double>& L);
// From a symmetric positive definite PRECISION matrix P = Sigma^-1: W lower triangular with W^T W = P (so |W dx|^2 is the
// quadratic form) and L = W^-1 (Sigma = L L^T), without inverting P.  false if P is not positive definite.
bool precision_factors(int n, const std::vector<double>& P, std::vector<double>& W, std::vector<double>& L);
// Inverse of a symmetric positive definite matrix through its Cholesky factor.
bool spd_inverse(int n, const std::vector<double>& P, std::vector<double>& S);
// Offset of element (row, col) in the pair-interleaved column layout the kernels stream
// (mvn_kernels.hip, "Packed factor access").
size_t packed_index(int R, int row, int col);
// mu / invdiag padded to 64 R, forward factor L_ij/L_ii and backward factor L_ij/L_jj packed.
void pack_factors(int n, int R, const std::vector<double>& L, std::vector<double>& mu_pad, const double* mu,
                  std::vector<double>& invdiag, std::vector<double>& Ft, std::vector<double>& Ut);

// W = L^-1 (row-major lower triangular, given) as the operand tiles k_wide.hip streams: row block ib (16 rows) holds the
// k tiles kt = 0 .. 4 (ib + 1) - 1 (4 columns each) at tile index 2 ib (ib + 1) + kt; a tile is 64 doubles in lane
// order, lane l = W[16 ib + (l & 15)][4 kt + (l >> 4)] (the A operand of v_mfma_f64_16x16x4_f64); zeros above the
// diagonal and beyond n.
// Wtb: the same for the transposed product y = W^T z (k_wide_grad.hip): row block ib holds the k tiles kt = 4 ib ..
// 4 NB - 1 at tile index 4 (ib NB - ib (ib - 1) / 2) + (kt - 4 ib), lane l = W[4 kt + (l >> 4)][16 ib + (l & 15)].
void pack_w_tiles(int n, const std::vector<double>& W, std::vector<double>& Wt, std::vector<double>& Wtb);
// W = L^-1, row-major lower triangular, long double accumulation.
void invert_factor(int n, const std::vector<double>& L, std::vector<double>& W);

// Tile stream of the row-split form (k_split.hip) for G row groups per chain tile.  The schedule itself is arithmetic
// (split_sched.hpp) shared with the device; here the tiles are packed group after group in the order of a group's stream, in
// pairs (lane l: its element of tile 2 i, then of tile 2 i + 1), plus one zero pair at the end; base[g] = first tile of group g.
constexpr int SPH_MAXSEG = 10;
struct SplitScheduleHost {
    int G = 0, nc = 0, NB = 0;
    std::vector<double> Ws;
    std::vector<int32_t> base;
};
void build_split_schedule(int n, int G, const std::vector<double>& W, SplitScheduleHost& out);

}  // namespace mcd
