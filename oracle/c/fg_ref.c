/* C/OpenMP restatement of the reference's per-iteration loop nests -- the CPU checker and
 * the "fibergen OpenMP path" stand-in for bench.py's cpu_baseline.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under fibergen_amd/ links or calls this.
 *
 * It keeps the reference's pass structure: a separate
 * polarisation sweep with schedule(dynamic) collapse(2) (F:18153), three divergence loop
 * nests with the reference's traversal orders (F:18864-18906), a separate 1/N scaling sweep
 * (F:18501-18506), sin/exp evaluated per frequency inside the Green-operator loop
 * (F:19873-19876), four strain loop nests (F:18627-18690), the eps += R sweep (F:20269) and
 * the norm sweep (F:10088-10138).  The FFT itself is supplied by the caller (pocketfft via
 * scipy.fft with `workers`), standing in for threaded FFTW.
 *
 * Arrays are float64, C order [comp][nx][ny][nz] (no z padding), components 11,22,33,23,13,12.
 * F: = /root/reference/src/fibergen.cpp @ 2024_08_07.  Build: gcc -O2 -fopenmp -ffp-contract=off.
 */
#include <complex.h>
#include <math.h>
#include <stddef.h>
#include <string.h>

#define IDX(i, j, k) (((size_t)(i) * ny + (j)) * nz + (k))

/* LinearIsotropicMaterialLaw::PK1  F:11375-11396 */
static void hooke6(const double* E, double mu, double lambda, double alpha, int accumulate, double* S) {
  const double two_mu = 2 * alpha * mu;
  const double lambda_tr_E = alpha * lambda * (E[0] + E[1] + E[2]);
  if (accumulate) {
    S[0] += E[0] * two_mu + lambda_tr_E; S[1] += E[1] * two_mu + lambda_tr_E; S[2] += E[2] * two_mu + lambda_tr_E;
    S[3] += E[3] * two_mu; S[4] += E[4] * two_mu; S[5] += E[5] * two_mu;
  } else {
    S[0] = E[0] * two_mu + lambda_tr_E; S[1] = E[1] * two_mu + lambda_tr_E; S[2] = E[2] * two_mu + lambda_tr_E;
    S[3] = E[3] * two_mu; S[4] = E[4] * two_mu; S[5] = E[5] * two_mu;
  }
}

static const int ROW[9] = {0, 1, 2, 1, 0, 0, 2, 2, 1}; /* F:13186-13188 */
static const int COL[9] = {0, 1, 2, 2, 2, 1, 1, 0, 0};

static void fix_dim(double* t) { t[6] = t[3]; t[7] = t[4]; t[8] = t[5]; }
static void fix_sym(double* t) {
  t[6] = t[3] = 0.5 * (t[3] + t[6]); t[7] = t[4] = 0.5 * (t[4] + t[7]); t[8] = t[5] = 0.5 * (t[5] + t[8]);
}
static double dot9(const double* A, const double* B) {
  return B[0] * A[0] + B[1] * A[1] + B[2] * A[2] + B[3] * A[3] + B[4] * A[4] + B[5] * A[5] + B[6] * A[6] + B[7] * A[7] +
         B[8] * A[8];
}

/* LaminateMixedMaterialLaw::solve_newton, DIM == 6  F:13157-13371 */
static void laminate_split(const double* Fbar, const double* n, double c1, double c2, double mu1, double l1, double mu2,
                           double l2, double eps_g, double eps_a, double* F1, double* F2) {
  double Fb[9], dF1[3][9], dF2[3][9], P1[9], P2[9], g[3], H[6], Hi[6], da[3], a[3], t;
  int i, k;
  for (i = 0; i < 6; i++) Fb[i] = Fbar[i];
  fix_dim(Fb);
  for (i = 0; i < 9; i++) F1[i] = F2[i] = Fb[i];
  for (k = 0; k < 3; k++) {
    for (i = 0; i < 9; i++) {
      const double rt = (ROW[i] == k) ? 1.0 : 0.0;
      dF1[k][i] = -c2 * rt * n[COL[i]];
      dF2[k][i] = c1 * rt * n[COL[i]];
    }
    fix_sym(dF1[k]);
    fix_sym(dF2[k]);
  }
  hooke6(F1, mu1, l1, 1.0, 0, P1); fix_dim(P1);
  hooke6(F2, mu2, l2, 1.0, 0, P2); fix_dim(P2);
  for (k = 0; k < 3; k++) g[k] = c1 * dot9(P1, dF1[k]) + c2 * dot9(P2, dF2[k]);
  t = 0; for (k = 0; k < 3; k++) t += g[k] * g[k];
  if (sqrt(t) <= eps_g) return;
  for (i = 0; i < 6; i++) {
    double dP1[9], dP2[9];
    const int kk = ROW[i], l = COL[i];
    hooke6(dF1[l], mu1, l1, 1.0, 0, dP1); fix_dim(dP1);
    hooke6(dF2[l], mu2, l2, 1.0, 0, dP2); fix_dim(dP2);
    H[i] = c1 * dot9(dP1, dF1[kk]) + c2 * dot9(dP2, dF2[kk]);
  }
  {
    const double det = H[0] * (H[1] * H[2] - H[3] * H[3]) - H[5] * (H[5] * H[2] - H[3] * H[4]) +
                       H[4] * (H[5] * H[3] - H[1] * H[4]);
    const double invdet = 1 / det;
    Hi[0] = (H[1] * H[2] - H[3] * H[3]) * invdet; Hi[1] = (H[0] * H[2] - H[4] * H[4]) * invdet;
    Hi[2] = (H[0] * H[1] - H[5] * H[5]) * invdet; Hi[3] = -(H[0] * H[3] - H[4] * H[5]) * invdet;
    Hi[4] = (H[5] * H[3] - H[4] * H[1]) * invdet; Hi[5] = -(H[5] * H[2] - H[3] * H[4]) * invdet;
  }
  da[0] = Hi[0] * g[0] + Hi[5] * g[1] + Hi[4] * g[2];
  da[1] = Hi[5] * g[0] + Hi[1] * g[1] + Hi[3] * g[2];
  da[2] = Hi[4] * g[0] + Hi[3] * g[1] + Hi[2] * g[2];
  t = 0; for (k = 0; k < 3; k++) t += da[k] * da[k];
  if (sqrt(t) <= eps_a) return;
  for (i = 0; i < 3; i++) a[i] = 0.0 - 1.0 * da[i];
  for (i = 0; i < 9; i++) {
    F1[i] -= c2 * a[ROW[i]] * n[COL[i]];
    F2[i] += c1 * a[ROW[i]] * n[COL[i]];
  }
  fix_sym(F1);
  fix_sym(F2);
}

/* _mat->PK1: Voigt F:12752-12761 (mixing 0) / laminate get_mix + PK1 F:13456-13558 (mixing 1); returns 1 on the
 * "only two phase mixtures" error */
static int pk1(const double* F, const double* phi, const double* nrm, int nph, const double* mu, const double* lambda,
               int mixing, double alpha, double eps_g, double eps_a, double* P) {
  int p;
  if (mixing == 0) {
    const double threshold = 10 * 2.220446049250313e-16;
    int any = 0;
    for (p = 0; p < nph; p++) {
      if (phi[p] <= threshold) continue;
      hooke6(F, mu[p], lambda[p], phi[p] * alpha, any, P);
      any = 1;
    }
    if (!any) memset(P, 0, 6 * sizeof(double));
    return 0;
  } else {
    int p1 = -1, p2 = -1;
    double c1 = 0, c2 = 0, F1[9], F2[9];
    for (p = 0; p < nph; p++) {
      const double f = phi[p];
      if (f == 0) continue;
      if (f == 1) { c1 = f; p1 = p; p2 = -1; break; }
      if (p1 < 0) { p1 = p; c1 = f; continue; }
      if (p2 < 0) { p2 = p; c2 = f; continue; }
      return 1;
    }
    if (p1 < 0) return 1;
    if (p2 < 0) { hooke6(F, mu[p1], lambda[p1], c1 * alpha, 0, P); return 0; }
    c2 = 1.0 - c1;
    laminate_split(F, nrm, c1, c2, mu[p1], lambda[p1], mu[p2], lambda[p2], eps_g, eps_a, F1, F2);
    hooke6(F1, mu[p1], lambda[p1], c1 * alpha, 0, P);
    hooke6(F2, mu[p2], lambda[p2], c2 * alpha, 1, P);
    return 0;
  }
}

/* calcStress  F:18134-18184 */
int ref_calc_stress(int nx, int ny, int nz, const double* eps, const double* phi, const double* normals, int nph,
                    const double* mu, const double* lambda, int mixing, double mu_0, double lambda_0, double alpha,
                    double eps_g, double eps_a, double* tau) {
  const size_t N = (size_t)nx * ny * nz;
  const double beta = -alpha * 2 * mu_0;
  const double gamma = -alpha * lambda_0;
  int err = 0;
#pragma omp parallel for schedule(dynamic) collapse(2) reduction(| : err)
  for (int i = 0; i < nx; i++)
    for (int j = 0; j < ny; j++)
      for (int k = 0; k < nz; k++) {
        const size_t o = IDX(i, j, k);
        double F[6], P[6], ph[16], nv[3] = {0, 0, 0};
        for (int c = 0; c < 6; c++) F[c] = eps[c * N + o];
        for (int p = 0; p < nph; p++) ph[p] = phi[p * N + o];
        if (normals) { nv[0] = normals[o]; nv[1] = normals[N + o]; nv[2] = normals[2 * N + o]; }
        err |= pk1(F, ph, nv, nph, mu, lambda, mixing, alpha, eps_g, eps_a, P);
        if (beta != 0) for (int c = 0; c < 6; c++) P[c] += beta * F[c];
        if (gamma != 0) {
          const double trF = F[0] + F[1] + F[2];
          for (int c = 0; c < 3; c++) P[c] += gamma * trF;
        }
        for (int c = 0; c < 6; c++) tau[c * N + o] = P[c];
      }
  return err;
}

/* meanPK1  F:12312-12351 (alpha / N, accumulate per thread, critical reduction) */
int ref_mean_stress(int nx, int ny, int nz, const double* eps, const double* phi, const double* normals, int nph,
                    const double* mu, const double* lambda, int mixing, double eps_g, double eps_a, double* S6) {
  const size_t N = (size_t)nx * ny * nz;
  const double alpha = 1.0 / (double)N;
  int err = 0;
  for (int c = 0; c < 6; c++) S6[c] = 0;
#pragma omp parallel
  {
    double SP[6] = {0, 0, 0, 0, 0, 0};
    int e = 0;
#pragma omp for schedule(dynamic) collapse(2)
    for (int i = 0; i < nx; i++)
      for (int j = 0; j < ny; j++)
        for (int k = 0; k < nz; k++) {
          const size_t o = IDX(i, j, k);
          double F[6], P[6], ph[16], nv[3] = {0, 0, 0};
          for (int c = 0; c < 6; c++) F[c] = eps[c * N + o];
          for (int p = 0; p < nph; p++) ph[p] = phi[p * N + o];
          if (normals) { nv[0] = normals[o]; nv[1] = normals[N + o]; nv[2] = normals[2 * N + o]; }
          e |= pk1(F, ph, nv, nph, mu, lambda, mixing, alpha, eps_g, eps_a, P);
          for (int c = 0; c < 6; c++) SP[c] += P[c];
        }
#pragma omp critical
    {
      for (int c = 0; c < 6; c++) S6[c] += SP[c];
      err |= e;
    }
  }
  return err;
}

/* divOperatorStaggered  F:18853-18908: three loop nests, inner loops along x, y, z respectively */
void ref_div(int nx, int ny, int nz, double dx, double dy, double dz, const double* x, double* y) {
  const size_t N = (size_t)nx * ny * nz;
  const double hx = nx / dx, hy = ny / dy, hz = nz / dz;
  const double *x0 = x, *x1 = x + N, *x2 = x + 2 * N, *x3 = x + 3 * N, *x4 = x + 4 * N, *x5 = x + 5 * N;
#pragma omp parallel
  {
#pragma omp for nowait schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int jj = 0; jj < ny; jj++) {
        double a0 = x0[IDX(nx - 1, jj, kk)];
        const int jf = (jj + 1) % ny, kf = (kk + 1) % nz;
        for (int ii = 0; ii < nx; ii++) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = x0[k];
          y[k] = (a1 - a0) * hx + (x5[IDX(ii, jf, kk)] - x5[k]) * hy + (x4[IDX(ii, jj, kf)] - x4[k]) * hz;
          a0 = a1;
        }
      }
#pragma omp for nowait schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = x1[IDX(ii, ny - 1, kk)];
        const int xf = (ii + 1) % nx, kf = (kk + 1) % nz;
        for (int jj = 0; jj < ny; jj++) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = x1[k];
          y[N + k] = (x5[IDX(xf, jj, kk)] - x5[k]) * hx + (a1 - a0) * hy + (x3[IDX(ii, jj, kf)] - x3[k]) * hz;
          a0 = a1;
        }
      }
#pragma omp for nowait schedule(static) collapse(2)
    for (int jj = 0; jj < ny; jj++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = x2[IDX(ii, jj, nz - 1)];
        const int xf = (ii + 1) % nx, jf = (jj + 1) % ny;
        for (int kk = 0; kk < nz; kk++) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = x2[k];
          y[2 * N + k] = (x4[IDX(xf, jj, kk)] - x4[k]) * hx + (x3[IDX(ii, jf, kk)] - x3[k]) * hy + (a1 - a0) * hz;
          a0 = a1;
        }
      }
  }
}

/* the 1/N sweep of fftVector  F:18501-18506 (complex data as interleaved doubles) */
void ref_scale(size_t n, double scale, double* y) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) y[i] *= scale;
}

/* G0OperatorFourierStaggered + General  F:19749-19755, F:19834-19927; fh = [3][nx][ny][nzc] complex, in place */
void ref_g0(int nx, int ny, int nz, double dx, double dy, double dz, double mu_0, double lambda_0, double alpha,
            double _Complex* fh) {
  const int nzc = nz / 2 + 1;
  const size_t F = (size_t)nx * ny * nzc;
  const double c10 = -alpha / (mu_0);
  const double c20 = -alpha / (mu_0 * (1 + mu_0 / (lambda_0 + mu_0)));
  const double h0 = dx / (2 * nx), h1 = dy / (2 * ny), h2 = dz / (2 * nz);
  const double xi0_0 = 2 * M_PI * h0 / (dx), xi1_0 = 2 * M_PI * h1 / (dy), xi2_0 = 2 * M_PI * h2 / (dz);
  const size_t ii_half = (nx & 1) == 0 ? (size_t)(nx / 2 - 1) : (size_t)(nx / 2);
  const size_t jj_half = (ny & 1) == 0 ? (size_t)(ny / 2 - 1) : (size_t)(ny / 2);
  const size_t kk_half = (nz & 1) == 0 ? (size_t)(nz / 2 - 1) : (size_t)(nz / 2);
  double _Complex *t0 = fh, *t1 = fh + F, *t2 = fh + 2 * F;
#pragma omp parallel for schedule(static)
  for (size_t ii = 0; ii < (size_t)nx; ii++) {
    const double xi0 = xi0_0 * ((ii <= ii_half) ? (double)ii : ((double)ii - (double)nx));
    const double kpm0 = sin(xi0) / h0;
    const double _Complex kp0 = kpm0 * cexp(I * xi0);
    const double _Complex km0 = -creal(kp0) + I * cimag(kp0);
    for (size_t jj = 0; jj < (size_t)ny; jj++) {
      const double xi1 = xi1_0 * ((jj <= jj_half) ? (double)jj : ((double)jj - (double)ny));
      const double kpm1 = sin(xi1) / h1;
      const double _Complex kp1 = kpm1 * cexp(I * xi1);
      const double _Complex km1 = -creal(kp1) + I * cimag(kp1);
      size_t k = (ii * ny + jj) * nzc;
      for (size_t kk = 0; kk < (size_t)nzc; kk++) {
        const double xi2 = xi2_0 * ((kk <= kk_half) ? (double)kk : ((double)kk - (double)nz));
        const double kpm2 = sin(xi2) / h2;
        const double _Complex kp2 = kpm2 * cexp(I * xi2);
        const double _Complex km2 = -creal(kp2) + I * cimag(kp2);
        const double norm_kp2 = kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2;
        const double c1 = c10 / (norm_kp2);
        const double c2 = c20 / (norm_kp2 * norm_kp2);
        const double _Complex c2_fkp = c2 * (t0[k] * kp0 + t1[k] * kp1 + t2[k] * kp2);
        const double _Complex e0 = c1 * t0[k] + c2_fkp * km0;
        const double _Complex e1 = c1 * t1[k] + c2_fkp * km1;
        const double _Complex e2 = c1 * t2[k] + c2_fkp * km2;
        t0[k] = e0; t1[k] = e1; t2[k] = e2;
        k++;
      }
    }
  }
  t0[0] = 0; t1[0] = 0; t2[0] = 0;
}

/* epsOperatorStaggered  F:18614-18692: shear nest, then three diagonal nests */
void ref_eps(int nx, int ny, int nz, double dx, double dy, double dz, const double* E, const double* x, double* y) {
  const size_t N = (size_t)nx * ny * nz;
  const double hx = nx / dx, hy = ny / dy, hz = nz / dz;
  const double *u0 = x, *u1 = x + N, *u2 = x + 2 * N;
#pragma omp parallel
  {
#pragma omp for schedule(static) collapse(2)
    for (int ii = 0; ii < nx; ii++)
      for (int jj = 0; jj < ny; jj++) {
        const int xb = (ii + nx - 1) % nx, yb = (jj + ny - 1) % ny;
        for (int kk = 0; kk < nz; kk++) {
          const int zb = (kk + nz - 1) % nz;
          const size_t k = IDX(ii, jj, kk);
          y[3 * N + k] = E[3] + 0.5 * ((u2[k] - u2[IDX(ii, yb, kk)]) * hy + (u1[k] - u1[IDX(ii, jj, zb)]) * hz);
          y[4 * N + k] = E[4] + 0.5 * ((u2[k] - u2[IDX(xb, jj, kk)]) * hx + (u0[k] - u0[IDX(ii, jj, zb)]) * hz);
          y[5 * N + k] = E[5] + 0.5 * ((u1[k] - u1[IDX(xb, jj, kk)]) * hx + (u0[k] - u0[IDX(ii, yb, kk)]) * hy);
        }
      }
#pragma omp for nowait schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int jj = 0; jj < ny; jj++) {
        double a0 = u0[IDX(0, jj, kk)];
        for (int ii = nx - 1; ii >= 0; ii--) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = u0[k];
          y[k] = E[0] + (a0 - a1) * hx;
          a0 = a1;
        }
      }
#pragma omp for nowait schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = u1[IDX(ii, 0, kk)];
        for (int jj = ny - 1; jj >= 0; jj--) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = u1[k];
          y[N + k] = E[1] + (a0 - a1) * hy;
          a0 = a1;
        }
      }
#pragma omp for nowait schedule(static) collapse(2)
    for (int jj = 0; jj < ny; jj++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = u2[IDX(ii, jj, 0)];
        for (int kk = nz - 1; kk >= 0; kk--) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = u2[k];
          y[2 * N + k] = E[2] + (a0 - a1) * hz;
          a0 = a1;
        }
      }
  }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * The two stencil operators with z innermost in EVERY loop nest (cpu_baseline's "tuned_loops" figure only).  The reference's
 * divOperatorStaggered / epsOperatorStaggered run their x- and y-difference nests with z OUTERMOST and x resp. y innermost
 * (F:18864-18887, F:18646-18675): every access of the inner loop is ny*nzp*8 resp. nzp*8 bytes from the previous one, and
 * under `omp for collapse(2)` neighbouring z -- adjacent doubles of ONE cache line -- belong to different threads (false sharing
 * on every store).  ref_div / ref_eps above keep those orders (they ARE the reference's OpenMP path, and why it stops scaling
 * at a few dozen threads); these two compute the same values element for element (same operands, same operation order per
 * element: bit-identical, tests/test_c_oracle.py) with unit-stride inner loops, to show what the traversal order costs. */
void ref_div_contig(int nx, int ny, int nz, double dx, double dy, double dz, const double* x, double* y) {
  const size_t N = (size_t)nx * ny * nz;
  const double hx = nx / dx, hy = ny / dy, hz = nz / dz;
  const double *x0 = x, *x1 = x + N, *x2 = x + 2 * N, *x3 = x + 3 * N, *x4 = x + 4 * N, *x5 = x + 5 * N;
#pragma omp parallel for schedule(static) collapse(2)
  for (int ii = 0; ii < nx; ii++)
    for (int jj = 0; jj < ny; jj++) {
      const int xf = (ii + 1) % nx, xb = (ii + nx - 1) % nx, jf = (jj + 1) % ny, jb = (jj + ny - 1) % ny;
      for (int kk = 0; kk < nz; kk++) {
        const int kf = (kk + 1) % nz, kb = (kk + nz - 1) % nz;
        const size_t k = IDX(ii, jj, kk);
        y[k] = (x0[k] - x0[IDX(xb, jj, kk)]) * hx + (x5[IDX(ii, jf, kk)] - x5[k]) * hy + (x4[IDX(ii, jj, kf)] - x4[k]) * hz;
        y[N + k] = (x5[IDX(xf, jj, kk)] - x5[k]) * hx + (x1[k] - x1[IDX(ii, jb, kk)]) * hy + (x3[IDX(ii, jj, kf)] - x3[k]) * hz;
        y[2 * N + k] = (x4[IDX(xf, jj, kk)] - x4[k]) * hx + (x3[IDX(ii, jf, kk)] - x3[k]) * hy + (x2[k] - x2[IDX(ii, jj, kb)]) * hz;
      }
    }
}

void ref_eps_contig(int nx, int ny, int nz, double dx, double dy, double dz, const double* E, const double* x, double* y) {
  const size_t N = (size_t)nx * ny * nz;
  const double hx = nx / dx, hy = ny / dy, hz = nz / dz;
  const double *u0 = x, *u1 = x + N, *u2 = x + 2 * N;
#pragma omp parallel for schedule(static) collapse(2)
  for (int ii = 0; ii < nx; ii++)
    for (int jj = 0; jj < ny; jj++) {
      const int xf = (ii + 1) % nx, xb = (ii + nx - 1) % nx, yf = (jj + 1) % ny, yb = (jj + ny - 1) % ny;
      for (int kk = 0; kk < nz; kk++) {
        const int zf = (kk + 1) % nz, zb = (kk + nz - 1) % nz;
        const size_t k = IDX(ii, jj, kk);
        y[3 * N + k] = E[3] + 0.5 * ((u2[k] - u2[IDX(ii, yb, kk)]) * hy + (u1[k] - u1[IDX(ii, jj, zb)]) * hz);
        y[4 * N + k] = E[4] + 0.5 * ((u2[k] - u2[IDX(xb, jj, kk)]) * hx + (u0[k] - u0[IDX(ii, jj, zb)]) * hz);
        y[5 * N + k] = E[5] + 0.5 * ((u1[k] - u1[IDX(xb, jj, kk)]) * hx + (u0[k] - u0[IDX(ii, yb, kk)]) * hy);
        y[k] = E[0] + (u0[IDX(xf, jj, kk)] - u0[k]) * hx;
        y[N + k] = E[1] + (u1[IDX(ii, yf, kk)] - u1[k]) * hy;
        y[2 * N + k] = E[2] + (u2[IDX(ii, jj, zf)] - u2[k]) * hz;
      }
    }
}

/* TensorField::add(R)  F:9841-9854 (called even when R == 0, F:20269) */
void ref_add(size_t N, const double* R, double* eps) {
#pragma omp parallel for schedule(static) collapse(2)
  for (int c = 0; c < 6; c++)
    for (size_t i = 0; i < N; i++) eps[c * N + i] += R[c];
}

/* component_dot / component_norm  F:10088-10138: per-component sum of squares / N, sqrt */
void ref_component_norm(size_t N, const double* eps, double* m6) {
  double a[6] = {0, 0, 0, 0, 0, 0};
#pragma omp parallel
  {
    double ap[6] = {0, 0, 0, 0, 0, 0};
#pragma omp for schedule(static)
    for (size_t i = 0; i < N; i++)
      for (int c = 0; c < 6; c++) ap[c] += eps[c * N + i] * eps[c * N + i];
#pragma omp critical
    for (int c = 0; c < 6; c++) a[c] += ap[c];
  }
  for (int c = 0; c < 6; c++) m6[c] = sqrt(a[c] / (double)N);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Vector routines of runCGElasticity  F:23153-23247 (the loop itself is driven from oracle/c_oracle.py: CRefCG), for
 * fields of `dim` components (6: strain fields, 3: the gradient fields of the scalar modes). */

/* innerProductL2  F:20871-20953 (a : (b - c), c != NULL) and F:20955-21038 (a : b): shear terms doubled for dim 6,
 * static schedule over the rows, the sum divided by the number of voxels */
double ref_inner_l2(int nx, int ny, int nz, int dim, const double* a, const double* b, const double* c) {
  const size_t N = (size_t)nx * ny * nz;
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static) collapse(2)
  for (int ii = 0; ii < nx; ii++)
    for (int jj = 0; jj < ny; jj++) {
      size_t k = IDX(ii, jj, 0);
      for (int kk = 0; kk < nz; kk++, k++) {
        if (dim == 6) {
          if (c)
            s += a[k] * (b[k] - c[k]) + a[N + k] * (b[N + k] - c[N + k]) + a[2 * N + k] * (b[2 * N + k] - c[2 * N + k]) +
                 2 * (a[3 * N + k] * (b[3 * N + k] - c[3 * N + k]) + a[4 * N + k] * (b[4 * N + k] - c[4 * N + k]) +
                      a[5 * N + k] * (b[5 * N + k] - c[5 * N + k]));
          else
            s += a[k] * b[k] + a[N + k] * b[N + k] + a[2 * N + k] * b[2 * N + k] +
                 2 * (a[3 * N + k] * b[3 * N + k] + a[4 * N + k] * b[4 * N + k] + a[5 * N + k] * b[5 * N + k]);
        } else {
          if (c) s += a[k] * (b[k] - c[k]) + a[N + k] * (b[N + k] - c[N + k]) + a[2 * N + k] * (b[2 * N + k] - c[2 * N + k]);
          else s += a[k] * b[k] + a[N + k] * b[N + k] + a[2 * N + k] * b[2 * N + k];
        }
      }
    }
  return s / (double)N;
}

/* TensorField::xpay  F:9819-9838: t = x + a y  (t may alias x or y) */
void ref_xpay(size_t n, const double* x, double a, const double* y, double* t) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) t[i] = x[i] + a * y[i];
}

/* TensorField::xpaymz  F:9993-10010: t = x + a (y - z) */
void ref_xpaymz(size_t n, const double* x, double a, const double* y, const double* z, double* t) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) t[i] = x[i] + a * (y[i] - z[i]);
}

/* TensorField::adjustResidual  F:10012-10022: r[j] += E[j] - z[j] */
void ref_adjust_residual(size_t N, int dim, const double* E, const double* z, double* r) {
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = 0; j < dim; j++)
    for (size_t i = 0; i < N; i++) r[j * N + i] += E[j] - z[j * N + i];
}

/* TensorField::setConstant(vector)  F:10044-10056 */
void ref_set_constant(size_t N, int dim, const double* E, double* t) {
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = 0; j < dim; j++)
    for (size_t i = 0; i < N; i++) t[j * N + i] = E[j];
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Scalar modes (mode = heat / porous, BASELINE config 5): 3-component gradient field g = E + grad T, one potential T.
 * Same conventions as above; the loop nests follow the reference's traversal orders. */

/* calcStress F:18134-18184 for dim 3 over VoigtMixedMaterialLaw::PK1 F:12752-12761 and
 * ScalarLinearIsotropicMaterialLaw::PK1 F:11182-11198: S_m (+)= E_m * ((phi * alpha) * mu), phases with phi <= 10 eps
 * skipped (F:12736); then tau += beta g, beta = -alpha 2 mu_0 (lambda_0 = 0 in these modes). */
void ref_calc_stress_scalar(int nx, int ny, int nz, const double* g, const double* phi, int nph, const double* mu, double mu_0,
                            double alpha, double* tau) {
  const size_t N = (size_t)nx * ny * nz;
  const double beta = -alpha * 2 * mu_0;
  const double thr = 10 * 2.220446049250313e-16;
#pragma omp parallel for schedule(dynamic) collapse(2)
  for (int i = 0; i < nx; i++)
    for (int j = 0; j < ny; j++)
      for (int k = 0; k < nz; k++) {
        const size_t o = IDX(i, j, k);
        double F[3], P[3] = {0, 0, 0};
        int first = 1;
        for (int c = 0; c < 3; c++) F[c] = g[c * N + o];
        for (int p = 0; p < nph; p++) {
          const double ph = phi[p * N + o];
          if (ph <= thr) continue;
          const double am = (ph * alpha) * mu[p];
          for (int c = 0; c < 3; c++) P[c] = first ? F[c] * am : P[c] + F[c] * am;
          first = 0;
        }
        if (beta != 0)
          for (int c = 0; c < 3; c++) P[c] += beta * F[c];
        for (int c = 0; c < 3; c++) tau[c * N + o] = P[c];
      }
}

/* divOperatorStaggeredHeat  F:18914-18975: backward differences accumulated into y[0] along x, then y, then z */
void ref_div_heat(int nx, int ny, int nz, double dx, double dy, double dz, const double* x, double* y) {
  const size_t N = (size_t)nx * ny * nz;
  const double hx = nx / dx, hy = ny / dy, hz = nz / dz;
  const double *x0p = x, *x1p = x + N, *x2p = x + 2 * N;
#pragma omp parallel
  {
#pragma omp for schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int jj = 0; jj < ny; jj++) {
        double a0 = x0p[IDX(nx - 1, jj, kk)];
        for (int ii = 0; ii < nx; ii++) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = x0p[k];
          y[k] = (a1 - a0) * hx;
          a0 = a1;
        }
      }
#pragma omp for schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = x1p[IDX(ii, ny - 1, kk)];
        for (int jj = 0; jj < ny; jj++) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = x1p[k];
          y[k] += (a1 - a0) * hy;
          a0 = a1;
        }
      }
#pragma omp for schedule(static) collapse(2)
    for (int jj = 0; jj < ny; jj++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = x2p[IDX(ii, jj, nz - 1)];
        for (int kk = 0; kk < nz; kk++) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = x2p[k];
          y[k] += (a1 - a0) * hz;
          a0 = a1;
        }
      }
  }
}

/* epsOperatorStaggeredHeat  F:18697-18760: forward differences of the potential plus E, lines walked backwards */
void ref_eps_heat(int nx, int ny, int nz, double dx, double dy, double dz, const double* E, const double* T, double* y) {
  const size_t N = (size_t)nx * ny * nz;
  const double hx = nx / dx, hy = ny / dy, hz = nz / dz;
#pragma omp parallel
  {
#pragma omp for nowait schedule(static) collapse(2)
    for (int jj = 0; jj < ny; jj++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = T[IDX(ii, jj, 0)];
        for (int kk = nz - 1; kk >= 0; kk--) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = T[k];
          y[2 * N + k] = E[2] + (a0 - a1) * hz;
          a0 = a1;
        }
      }
#pragma omp for nowait schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int ii = 0; ii < nx; ii++) {
        double a0 = T[IDX(ii, 0, kk)];
        for (int jj = ny - 1; jj >= 0; jj--) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = T[k];
          y[N + k] = E[1] + (a0 - a1) * hy;
          a0 = a1;
        }
      }
#pragma omp barrier
#pragma omp for nowait schedule(static) collapse(2)
    for (int kk = 0; kk < nz; kk++)
      for (int jj = 0; jj < ny; jj++) {
        double a0 = T[IDX(0, jj, kk)];
        for (int ii = nx - 1; ii >= 0; ii--) {
          const size_t k = IDX(ii, jj, kk);
          const double a1 = T[k];
          y[k] = E[0] + (a0 - a1) * hx;
          a0 = a1;
        }
      }
  }
}

/* G0OperatorFourierStaggeredHeat + GeneralHeat  F:19758-19823: c1 = c10 / |k|^2, c10 = -alpha / (2 mu_0); th = [nx][ny][nzc]
 * complex, in place; zero mode := 0 */
void ref_g0_heat(int nx, int ny, int nz, double dx, double dy, double dz, double mu_0, double alpha, double _Complex* th) {
  const int nzc = nz / 2 + 1;
  const double c10 = -alpha / (2 * mu_0);
  const double h0 = dx / (2 * nx), h1 = dy / (2 * ny), h2 = dz / (2 * nz);
  const double xi0_0 = 2 * M_PI * h0 / (dx), xi1_0 = 2 * M_PI * h1 / (dy), xi2_0 = 2 * M_PI * h2 / (dz);
  const size_t ii_half = (nx & 1) == 0 ? (size_t)(nx / 2 - 1) : (size_t)(nx / 2);
  const size_t jj_half = (ny & 1) == 0 ? (size_t)(ny / 2 - 1) : (size_t)(ny / 2);
  const size_t kk_half = (nz & 1) == 0 ? (size_t)(nz / 2 - 1) : (size_t)(nz / 2);
#pragma omp parallel for schedule(static)
  for (size_t ii = 0; ii < (size_t)nx; ii++) {
    const double xi0 = xi0_0 * ((ii <= ii_half) ? (double)ii : ((double)ii - (double)nx));
    const double kpm0 = sin(xi0) / h0;
    for (size_t jj = 0; jj < (size_t)ny; jj++) {
      const double xi1 = xi1_0 * ((jj <= jj_half) ? (double)jj : ((double)jj - (double)ny));
      const double kpm1 = sin(xi1) / h1;
      size_t k = (ii * ny + jj) * nzc;
      for (size_t kk = 0; kk < (size_t)nzc; kk++) {
        const double xi2 = xi2_0 * ((kk <= kk_half) ? (double)kk : ((double)kk - (double)nz));
        const double kpm2 = sin(xi2) / h2;
        const double norm_kp2 = kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2;
        const double c1 = c10 / (norm_kp2);
        th[k] = c1 * th[k];
        k++;
      }
    }
  }
  th[0] = 0;
}

int ref_max_threads(void);
#ifdef _OPENMP
#include <omp.h>
int ref_max_threads(void) { return omp_get_max_threads(); }
void ref_set_threads(int n) { omp_set_num_threads(n); }
#else
int ref_max_threads(void) { return 1; }
void ref_set_threads(int n) { (void)n; }
#endif
