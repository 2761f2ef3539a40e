// Per-voxel / per-frequency arithmetic of the Lippmann-Schwinger stages,
// shared by the HIP kernels (fg_kernels.hip) and the host emulation in tests.
//
// Operation order follows the reference exactly (citations F: =
// fibergen.cpp @ 2024_08_07); the library is built with -ffp-contract=off so
// these stages are bit-reproducible against a plain C evaluation.
#pragma once

#include <cmath>

#include "fg_common.h"

namespace fg {

constexpr int kMaxPhases = 8;

struct PhaseTable {
  int n;
  double mu[kMaxPhases];
  double lambda[kMaxPhases];
};

enum MixingRule { kMixVoigt = 0, kMixLaminate = 1 };

// LinearIsotropicMaterialLaw::PK1  F:11375-11396  (gamma = accumulate)
FG_HD void hooke6(const double* E, double mu, double lambda, double alpha, bool accumulate, double* S) {
  const double two_mu = 2 * alpha * mu;
  const double lambda_tr_E = alpha * lambda * (E[0] + E[1] + E[2]);
  if (accumulate) {
    S[0] += E[0] * two_mu + lambda_tr_E;
    S[1] += E[1] * two_mu + lambda_tr_E;
    S[2] += E[2] * two_mu + lambda_tr_E;
    S[3] += E[3] * two_mu;
    S[4] += E[4] * two_mu;
    S[5] += E[5] * two_mu;
  } else {
    S[0] = E[0] * two_mu + lambda_tr_E;
    S[1] = E[1] * two_mu + lambda_tr_E;
    S[2] = E[2] * two_mu + lambda_tr_E;
    S[3] = E[3] * two_mu;
    S[4] = E[4] * two_mu;
    S[5] = E[5] * two_mu;
  }
}

// VoigtMixedMaterialLaw::PK1  F:12752-12761 ; threshold 10*eps  F:12736
template <int NPH>
FG_HD void pk1_voigt(const double* F, const double* phi, const PhaseTable& pt, double alpha, bool gamma, double* P) {
  const double threshold = 10 * 2.220446049250313e-16;
  bool any = gamma;
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
    if (p >= pt.n) break;
    if (phi[p] <= threshold) continue;
    hooke6(F, pt.mu[p], pt.lambda[p], phi[p] * alpha, any, P);
    any = true;
  }
  if (!any) {
    for (int i = 0; i < 6; ++i) P[i] = 0.0;  // reference leaves P untouched; only padding gets here
  }
}

// 9-component index maps  F:13186-13188  (11,22,33,23,13,12,32,31,21)
FG_HD int lam_row(int i) {
  const int row[9] = {0, 1, 2, 1, 0, 0, 2, 2, 1};
  return row[i];
}
FG_HD int lam_col(int i) {
  const int col[9] = {0, 1, 2, 2, 2, 1, 1, 0, 0};
  return col[i];
}

FG_HD void fix_dim9(double* t) {  // F:12115-12125
  t[6] = t[3];
  t[7] = t[4];
  t[8] = t[5];
}
FG_HD void fix_sym9(double* t) {  // F:12128-12138
  t[6] = t[3] = 0.5 * (t[3] + t[6]);
  t[7] = t[4] = 0.5 * (t[4] + t[7]);
  t[8] = t[5] = 0.5 * (t[5] + t[8]);
}
FG_HD double dot9(const double* A, const double* B) {  // Tensor3x3::dot  F:9332-9335
  return B[0] * A[0] + B[1] * A[1] + B[2] * A[2] + B[3] * A[3] + B[4] * A[4] + B[5] * A[5] + B[6] * A[6] +
         B[7] * A[7] + B[8] * A[8];
}

// LaminateMixedMaterialLaw::solve_newton, DIM == 6 path  F:13157-13371:
// one Newton step from a = 0 for the jump a (x) n, then return.
FG_HD void laminate_split(const double* Fbar6, const double* n, double c1, double c2, double mu1, double lambda1,
                          double mu2, double lambda2, double eps_g, double eps_a, double* F1, double* F2) {
  double Fb[9];
  for (int i = 0; i < 6; ++i) Fb[i] = Fbar6[i];
  fix_dim9(Fb);
  for (int i = 0; i < 9; ++i) F1[i] = F2[i] = Fb[i];

  double dF1[3][9], dF2[3][9];
  for (int k = 0; k < 3; ++k) {
    for (int i = 0; i < 9; ++i) {
      const double rt = (lam_row(i) == k) ? 1.0 : 0.0;  // RT = identity  F:13176-13183
      dF1[k][i] = -c2 * rt * n[lam_col(i)];
      dF2[k][i] = c1 * rt * n[lam_col(i)];
    }
    fix_sym9(dF1[k]);
    fix_sym9(dF2[k]);
  }

  double P1[9], P2[9], g[3];
  hooke6(F1, mu1, lambda1, 1.0, false, P1);
  fix_dim9(P1);
  hooke6(F2, mu2, lambda2, 1.0, false, P2);
  fix_dim9(P2);
  for (int k = 0; k < 3; ++k) g[k] = c1 * dot9(P1, dF1[k]) + c2 * dot9(P2, dF2[k]);
  double t = 0.0;
  for (int k = 0; k < 3; ++k) t += g[k] * g[k];
  const double g_norm = sqrt(t);
  if (g_norm <= eps_g) return;  // F:13262

  double H[6];
  for (int i = 0; i < 6; ++i) {
    const int k = lam_row(i), l = lam_col(i);
    double dP1[9], dP2[9];
    hooke6(dF1[l], mu1, lambda1, 1.0, false, dP1);
    fix_dim9(dP1);
    hooke6(dF2[l], mu2, lambda2, 1.0, false, dP2);
    fix_dim9(dP2);
    H[i] = c1 * dot9(dP1, dF1[k]) + c2 * dot9(dP2, dF2[k]);
  }
  // SymTensor3x3::det / inv  F:9483-9488, F:9373-9382
  const double det = H[0] * (H[1] * H[2] - H[3] * H[3]) - H[5] * (H[5] * H[2] - H[3] * H[4]) +
                     H[4] * (H[5] * H[3] - H[1] * H[4]);
  const double invdet = 1 / det;
  double Hi[6];
  Hi[0] = (H[1] * H[2] - H[3] * H[3]) * invdet;
  Hi[1] = (H[0] * H[2] - H[4] * H[4]) * invdet;
  Hi[2] = (H[0] * H[1] - H[5] * H[5]) * invdet;
  Hi[3] = -(H[0] * H[3] - H[4] * H[5]) * invdet;
  Hi[4] = (H[5] * H[3] - H[4] * H[1]) * invdet;
  Hi[5] = -(H[5] * H[2] - H[3] * H[4]) * invdet;
  double da[3];  // Tensor3::mult  F:9516-9521
  da[0] = Hi[0] * g[0] + Hi[5] * g[1] + Hi[4] * g[2];
  da[1] = Hi[5] * g[0] + Hi[1] * g[1] + Hi[3] * g[2];
  da[2] = Hi[4] * g[0] + Hi[3] * g[1] + Hi[2] * g[2];
  t = 0.0;
  for (int k = 0; k < 3; ++k) t += da[k] * da[k];
  const double da_norm = sqrt(t);
  if (da_norm <= eps_a) return;  // F:13305

  double a[3];
  for (int i = 0; i < 3; ++i) a[i] = 0.0 - 1.0 * da[i];  // a_next = a - t*da  F:13338-13340
  for (int i = 0; i < 9; ++i) {                           // F:13348-13351
    F1[i] -= c2 * a[lam_row(i)] * n[lam_col(i)];
    F2[i] += c1 * a[lam_row(i)] * n[lam_col(i)];
  }
  fix_sym9(F1);
  fix_sym9(F2);
}

struct LaminateMix {
  int p1, p2;     // phase indices, -1 = none
  double c1, c2;
};

// get_mix phase selection  F:13461-13480 (without the Newton solve)
// returns 0 ok, 1 = more than two phases / no phase (the reference throws)
template <int NPH>
FG_HD int laminate_select(const double* phi, int nphase, LaminateMix& m) {
  m.p1 = m.p2 = -1;
  m.c1 = m.c2 = 0.0;
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
    if (p >= nphase) break;
    const double f = phi[p];
    if (f == 0) continue;
    if (f == 1) {
      m.c1 = f;
      m.p1 = p;
      m.p2 = -1;
      return 0;
    }
    if (m.p1 < 0) { m.p1 = p; m.c1 = f; continue; }
    if (m.p2 < 0) { m.p2 = p; m.c2 = f; continue; }
    return 1;
  }
  if (m.p1 < 0) return 1;
  if (m.p2 >= 0) m.c2 = 1.0 - m.c1;  // F:13523
  return 0;
}

// LaminateMixedMaterialLaw::PK1  F:13543-13558
template <int NPH>
FG_HD int pk1_laminate(const double* F, const double* phi, const double* normal, const PhaseTable& pt, double alpha,
                       bool gamma, double eps_g, double eps_a, double* P) {
  LaminateMix m;
  if (laminate_select<NPH>(phi, pt.n, m) != 0) {
    if (!gamma) for (int i = 0; i < 6; ++i) P[i] = 0.0;
    return 1;
  }
  if (m.p2 < 0) {
    hooke6(F, pt.mu[m.p1], pt.lambda[m.p1], m.c1 * alpha, gamma, P);
    return 0;
  }
  double F1[9], F2[9];
  laminate_split(F, normal, m.c1, m.c2, pt.mu[m.p1], pt.lambda[m.p1], pt.mu[m.p2], pt.lambda[m.p2], eps_g, eps_a,
                 F1, F2);
  hooke6(F1, pt.mu[m.p1], pt.lambda[m.p1], m.c1 * alpha, gamma, P);
  hooke6(F2, pt.mu[m.p2], pt.lambda[m.p2], m.c2 * alpha, true, P);
  return 0;
}

struct StressParams {
  PhaseTable pt;
  int mixing;
  double mu_0, lambda_0, alpha;
  double eps_g, eps_a;  // laminate tolerances  F:13110-13111
};

// calcStress voxel body  F:18156-18176 : P = PK1(F) + beta F + gamma tr(F) I
template <int NPH>
FG_HD int stress_voxel(const double* F, const double* phi, const double* normal, const StressParams& sp, double* P) {
  int err = 0;
  if (sp.mixing == kMixLaminate) err = pk1_laminate<NPH>(F, phi, normal, sp.pt, sp.alpha, false, sp.eps_g, sp.eps_a, P);
  else pk1_voigt<NPH>(F, phi, sp.pt, sp.alpha, false, P);
  const double beta = -sp.alpha * 2 * sp.mu_0;
  const double gamma = -sp.alpha * sp.lambda_0;
  if (beta != 0) {
    for (int k = 0; k < 6; ++k) P[k] += beta * F[k];
  }
  if (gamma != 0) {
    const double trF = F[0] + F[1] + F[2];
    for (int k = 0; k < 3; ++k) P[k] += gamma * trF;
  }
  return err;
}

// Single components of the Voigt-mixed polarisation tau = P(F) + beta F + gamma tr(F) I, evaluated
// with exactly the operations stress_voxel performs for that component (the components of
// pk1_voigt/hooke6 are independent), so that a kernel which needs only tau_c at a neighbour voxel
// reproduces the stored field bit for bit.   F:12752-12761, F:11375-11396, F:18162-18172
template <int NPH>
FG_HD double voigt_tau_normal(double Ec, double E0, double E1, double E2, const double* phi, const StressParams& sp) {
  const double threshold = 10 * 2.220446049250313e-16;
  double P = 0.0;
  bool any = false;
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
    if (p >= sp.pt.n) break;
    if (phi[p] <= threshold) continue;
    const double a = phi[p] * sp.alpha;
    const double two_mu = 2 * a * sp.pt.mu[p];
    const double lambda_tr_E = a * sp.pt.lambda[p] * (E0 + E1 + E2);
    if (any) P += Ec * two_mu + lambda_tr_E;
    else P = Ec * two_mu + lambda_tr_E;
    any = true;
  }
  const double beta = -sp.alpha * 2 * sp.mu_0;
  const double gamma = -sp.alpha * sp.lambda_0;
  if (beta != 0) P += beta * Ec;
  if (gamma != 0) P += gamma * (E0 + E1 + E2);
  return P;
}

template <int NPH>
FG_HD double voigt_tau_shear(double Ec, const double* phi, const StressParams& sp) {
  const double threshold = 10 * 2.220446049250313e-16;
  double P = 0.0;
  bool any = false;
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
    if (p >= sp.pt.n) break;
    if (phi[p] <= threshold) continue;
    const double two_mu = 2 * (phi[p] * sp.alpha) * sp.pt.mu[p];
    if (any) P += Ec * two_mu;
    else P = Ec * two_mu;
    any = true;
  }
  const double beta = -sp.alpha * 2 * sp.mu_0;
  if (beta != 0) P += beta * Ec;
  return P;
}

// Tangent spectrum of one voxel for the reference-material scan
// (getRefMaterial/eig  F:12153-12236, F:12472-12559).  For isotropic phases the
// Voigt tangent (F:12763-12771) and the laminate tangent="approx" (F:13611-13624)
// are isotropic: diag block 2mu I + lam 11^T, shear block 2mu I, eigenvalues
// {2mu (x5), 2mu + 3 lam}.  (The reference obtains them with LAPACK dsyev.)
template <int NPH>
FG_HD int tangent_eigs(const double* phi, const PhaseTable& pt, int mixing, double* emin, double* emax) {
  double two_mu = 0.0, lam = 0.0;
  if (mixing == kMixLaminate) {
    LaminateMix m;
    if (laminate_select<NPH>(phi, pt.n, m) != 0) return 1;
    two_mu = 2 * m.c1 * pt.mu[m.p1];
    lam = m.c1 * pt.lambda[m.p1];
    if (m.p2 >= 0) {
      two_mu += 2 * m.c2 * pt.mu[m.p2];
      lam += m.c2 * pt.lambda[m.p2];
    }
  } else {
    const double threshold = 10 * 2.220446049250313e-16;
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      if (p >= pt.n) break;
      if (phi[p] <= threshold) continue;
      two_mu += 2 * phi[p] * pt.mu[p];
      lam += phi[p] * pt.lambda[p];
    }
  }
  const double e1 = two_mu, e2 = two_mu + 3 * lam;
  *emin = e1 < e2 ? e1 : e2;
  *emax = e1 < e2 ? e2 : e1;
  return 0;
}

// G0OperatorFourierStaggeredGeneral frequency body  F:19873-19913
//   kpm_a = sin(xi_a)/h_a, kp_a = kpm_a e^{i xi_a}, km_a = (-Re kp_a, Im kp_a)
FG_HD void g0_point(cplx t0, cplx t1, cplx t2, double kpm0, double kpm1, double kpm2, cplx kp0, cplx kp1, cplx kp2,
                    double c10, double c20, cplx* e0, cplx* e1, cplx* e2) {
  const double norm_kp2 = kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2;
  const double c1 = c10 / norm_kp2;
  const double c2 = c20 / (norm_kp2 * norm_kp2);
  const cplx s = cadd(cadd(cmul(t0, kp0), cmul(t1, kp1)), cmul(t2, kp2));
  const cplx c2_fkp = cscale(c2, s);
  const cplx km0 = cmake(-kp0.re, kp0.im), km1 = cmake(-kp1.re, kp1.im), km2 = cmake(-kp2.re, kp2.im);
  *e0 = cadd(cscale(c1, t0), cmul(c2_fkp, km0));
  *e1 = cadd(cscale(c1, t1), cmul(c2_fkp, km1));
  *e2 = cadd(cscale(c1, t2), cmul(c2_fkp, km2));
}

// The same body with one division: c1 = c10 / |k|^2, c2 = c20 / |k|^4 from a single reciprocal.  Used by the
// fused FFT pass, which has no bit-exactness contract (the FFT itself replaces FFTW); differs from g0_point by a few ulp.
FG_HD void g0_point_rcp(cplx t0, cplx t1, cplx t2, double kpm0, double kpm1, double kpm2, cplx kp0, cplx kp1, cplx kp2,
                        double c10, double c20, cplx* e0, cplx* e1, cplx* e2) {
  const double inv = 1.0 / (kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2);
  const double c1 = c10 * inv;
  const double c2 = c20 * inv * inv;
  const cplx s = cadd(cadd(cmul(t0, kp0), cmul(t1, kp1)), cmul(t2, kp2));
  const cplx c2_fkp = cscale(c2, s);
  *e0 = cadd(cscale(c1, t0), cmul(c2_fkp, cmake(-kp0.re, kp0.im)));
  *e1 = cadd(cscale(c1, t1), cmul(c2_fkp, cmake(-kp1.re, kp1.im)));
  *e2 = cadd(cscale(c1, t2), cmul(c2_fkp, cmake(-kp2.re, kp2.im)));
}

}  // namespace fg
