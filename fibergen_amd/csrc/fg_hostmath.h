// Small dense host-side algebra of the solver driver: Voigt contractions
// (F:494-598) and the pseudo inverse needed by the boundary-condition projector
// (F:20599-20665).  The reference calls LAPACK gesvd on the symmetric 9x9 matrix
// Q:C0:Q; LAPACK is not available here, a cyclic Jacobi eigen-decomposition gives
// the same Moore-Penrose inverse for a symmetric matrix.
#pragma once

#include <cmath>
#include <cstring>

namespace fg {
namespace hostmath {

struct Mat6 {
  double a[6][6];
};

inline Mat6 mat6_zero() {
  Mat6 m;
  std::memset(&m, 0, sizeof(m));
  return m;
}

// Voigt::Id4(6)  F:501-512
inline Mat6 voigt_id4() {
  Mat6 m = mat6_zero();
  for (int i = 0; i < 3; ++i) m.a[i][i] = 1.0;
  for (int i = 3; i < 6; ++i) m.a[i][i] = 0.5;
  return m;
}

// Voigt::II4(6)  F:517-525
inline Mat6 voigt_ii4() {
  Mat6 m = mat6_zero();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m.a[i][j] = 1.0;
  return m;
}

// Voigt::dyad4(M, v): shear entries of v doubled  F:563-575
inline void voigt_mv(const Mat6& M, const double* v, double* out) {
  double vc[6];
  for (int i = 0; i < 6; ++i) vc[i] = v[i];
  vc[3] *= 2;
  vc[4] *= 2;
  vc[5] *= 2;
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int j = 0; j < 6; ++j) s += M.a[i][j] * vc[j];
    out[i] = s;
  }
}

// Voigt::dyad4(A, B) column by column  F:582-597
inline Mat6 voigt_mm(const Mat6& A, const Mat6& B) {
  Mat6 C;
  for (int c = 0; c < 6; ++c) {
    double col[6], out[6];
    for (int i = 0; i < 6; ++i) col[i] = B.a[i][c];
    voigt_mv(A, col, out);
    for (int i = 0; i < 6; ++i) C.a[i][c] = out[i];
  }
  return C;
}

// Voigt::norm_2  F:530-537
inline double voigt_norm2(const double* v) {
  double s = 0.0;
  for (int i = 0; i < 6; ++i) s += v[i] * v[i];
  return std::sqrt(s + v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
}

inline double frobenius(const Mat6& m) {
  double s = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) s += m.a[i][j] * m.a[i][j];
  return std::sqrt(s);
}

inline double norm2(const double* v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return std::sqrt(s);
}

// Symmetric N x N eigen-decomposition by cyclic Jacobi: A = V diag(w) V^T.
template <int N>
inline void jacobi_eig(double A[N][N], double V[N][N], double w[N]) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < N; ++i)
      for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
    if (off == 0.0) break;
    for (int p = 0; p < N; ++p) {
      for (int q = p + 1; q < N; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
    }
  }
  for (int i = 0; i < N; ++i) w[i] = A[i][i];
}

// Moore-Penrose inverse of the symmetric 6x6 Voigt operator Q:C0:Q through its
// 9x9 extension, then folded back to 6x6 -- setBCProjector  F:20621-20661.
inline Mat6 bc_pseudo_inverse(const Mat6& QC0Q) {
  double A[9][9], V[9][9], w[9];
  for (int i = 0; i < 9; ++i)
    for (int j = i; j < 9; ++j) A[j][i] = A[i][j] = QC0Q.a[i < 6 ? i : i - 3][j < 6 ? j : j - 3];
  jacobi_eig<9>(A, V, w);
  double snorm = 0.0;
  for (int i = 0; i < 9; ++i) snorm += w[i] * w[i];
  const double alpha = std::sqrt(2.220446049250313e-16) * std::sqrt(snorm);
  double M[9][9];
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      double s = 0.0;
      for (int k = 0; k < 9; ++k)
        if (std::fabs(w[k]) > alpha) s += V[i][k] * V[j][k] / w[k];
      M[i][j] = s;
    }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 6; ++j) M[j][3 + i] = 0.5 * (M[j][3 + i] + M[j][6 + i]);
    for (int j = 0; j < 6; ++j) M[3 + i][j] = 0.5 * (M[3 + i][j] + M[6 + i][j]);
  }
  Mat6 out;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) out.a[i][j] = M[i][j];
  return out;
}

}  // namespace hostmath
}  // namespace fg
