// host_factor.h -- one-time host preparation of the likelihood operands (internal).
#pragma once
#include <cstddef>
#include <vector>

namespace mcd {

// Lower Cholesky factor of a symmetric matrix (row-major n x n).  false if not positive definite.
bool cholesky_lower(int n, const std::vector<double>& A, std::vector<double>& L);
// Inverse of a symmetric positive definite matrix through its Cholesky factor.
bool spd_inverse(int n, const std::vector<double>& P, std::vector<double>& S);
// Offset of element (row, col) in the pair-interleaved column layout the kernels stream
// (mvn_kernels.hip, "Packed factor access").
size_t packed_index(int R, int row, int col);
// mu / invdiag padded to 64 R, forward factor L_ij/L_ii and backward factor L_ij/L_jj packed.
void pack_factors(int n, int R, const std::vector<double>& L, std::vector<double>& mu_pad, const double* mu,
                  std::vector<double>& invdiag, std::vector<double>& Ft, std::vector<double>& Ut);

}  // namespace mcd
