// Fast variants of the sweeps of the displacement-based loop: per-voxel effective moduli are
// precomputed (A = sum_p phi_p 2 mu_p, B = sum_p phi_p lambda_p, Voigt mixing F:12752-12761), so the
// polarisation of a voxel is  tau = (A - 2 mu0) eps + (B - lambda0) tr(eps) I  -- two or three flops per
// component instead of the per-phase accumulation, and this translation unit is compiled with FMA
// contraction.  Results agree with the exact-order kernels of fg_kernels.hip to rounding (~1e-16
// relative per operation, asserted to 1e-12 on fields); the exact kernels remain available (u_loop = 1).
#include "fg_kernels.h"

#include "fg_hip_util.h"
#include "fg_kernels_common.h"

namespace fg {

namespace {

__global__ __launch_bounds__(kBlock) void k_u_fast(Grid g, double beta, double gamma, FieldPtrs<3> u, FieldPtrs<2> mod,
                                                   FieldPtrs<3> fo, Vec6 E, double* partial, int ry) {
  __shared__ double smem[4 * 6];
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx = (run.first + it * run.stride) * kBlock + threadIdx.x;
    if (pidx >= npairs) continue;
    const PairPos p = pair_pos_tiled(pidx, g, ry);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    const long xf = (p.i + 1 == g.nx ? -(long)(g.nx - 1) : 1L) * g.nyzp;
    const long xb = (p.i == 0 ? (long)(g.nx - 1) : -1L) * g.nyzp;
    const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
    const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
    const long ro = p.off - p.k;
    const int k = p.k;
    const int kb = k == 0 ? g.nz - 1 : k - 1;
    const int kf2 = (k + 2 >= g.nz) ? k + 2 - g.nz : k + 2;
#define FG_ROW(a, off, m1, p2) load_row(a, ro + (off), k, kb, kf2, second, m1, p2)
    const Row4 U0c = FG_ROW(u.p[0], 0, true, true), U0xf = FG_ROW(u.p[0], xf, true, false);
    const Row4 U0yb = FG_ROW(u.p[0], yb, false, false), U0xb = FG_ROW(u.p[0], xb, false, false);
    const Row4 U0xfyb = FG_ROW(u.p[0], xf + yb, false, false), U0yf = FG_ROW(u.p[0], yf, false, false);
    const Row4 U1c = FG_ROW(u.p[1], 0, true, true), U1yf = FG_ROW(u.p[1], yf, true, false);
    const Row4 U1xb = FG_ROW(u.p[1], xb, false, false), U1xbyf = FG_ROW(u.p[1], xb + yf, false, false);
    const Row4 U1yb = FG_ROW(u.p[1], yb, false, false), U1xf = FG_ROW(u.p[1], xf, false, false);
    const Row4 U2c = FG_ROW(u.p[2], 0, true, true), U2yb = FG_ROW(u.p[2], yb, false, true);
    const Row4 U2xb = FG_ROW(u.p[2], xb, false, true), U2xf = FG_ROW(u.p[2], xf, false, false);
    const Row4 U2yf = FG_ROW(u.p[2], yf, false, false);
#undef FG_ROW
    // effective moduli rows: A = sum_p phi_p 2 mu_p, B = sum_p phi_p lambda_p (precomputed per voxel)
    const Row4 Ac = load_row(mod.p[0], ro, k, kb, kf2, second, true, true), Bc = load_row(mod.p[1], ro, k, kb, kf2, second, true, true);
    const Row4 Axb = load_row(mod.p[0], ro + xb, k, kb, kf2, second, false, false);
    const Row4 Bxb = load_row(mod.p[1], ro + xb, k, kb, kf2, second, false, false);
    const Row4 Axf = load_row(mod.p[0], ro + xf, k, kb, kf2, second, false, false);
    const Row4 Ayb = load_row(mod.p[0], ro + yb, k, kb, kf2, second, false, false);
    const Row4 Byb = load_row(mod.p[1], ro + yb, k, kb, kf2, second, false, false);
    const Row4 Ayf = load_row(mod.p[0], ro + yf, k, kb, kf2, second, false, false);

    double fout[2][3], eout[2][6];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int i0 = s + 1;  // index of this voxel's own z position in the row vectors
      // strain at the voxel  (F:18632-18686)
      const double e0 = E.v[0] + (U0xf.v[i0] - U0c.v[i0]) * hx;
      const double e1 = E.v[1] + (U1yf.v[i0] - U1c.v[i0]) * hy;
      const double e2 = E.v[2] + (U2c.v[i0 + 1] - U2c.v[i0]) * hz;
      const double e3 = E.v[3] + 0.5 * ((U2c.v[i0] - U2yb.v[i0]) * hy + (U1c.v[i0] - U1c.v[i0 - 1]) * hz);
      const double e4 = E.v[4] + 0.5 * ((U2c.v[i0] - U2xb.v[i0]) * hx + (U0c.v[i0] - U0c.v[i0 - 1]) * hz);
      const double e5 = E.v[5] + 0.5 * ((U1c.v[i0] - U1xb.v[i0]) * hx + (U0c.v[i0] - U0yb.v[i0]) * hy);
      eout[s][0] = e0; eout[s][1] = e1; eout[s][2] = e2; eout[s][3] = e3; eout[s][4] = e4; eout[s][5] = e5;
      // strains of the six neighbours, only the components their tau needs
      const double e0xb = E.v[0] + (U0c.v[i0] - U0xb.v[i0]) * hx;
      const double e1xb = E.v[1] + (U1xbyf.v[i0] - U1xb.v[i0]) * hy;
      const double e2xb = E.v[2] + (U2xb.v[i0 + 1] - U2xb.v[i0]) * hz;
      const double e0yb = E.v[0] + (U0xfyb.v[i0] - U0yb.v[i0]) * hx;
      const double e1yb = E.v[1] + (U1c.v[i0] - U1yb.v[i0]) * hy;
      const double e2yb = E.v[2] + (U2yb.v[i0 + 1] - U2yb.v[i0]) * hz;
      const double e0zb = E.v[0] + (U0xf.v[i0 - 1] - U0c.v[i0 - 1]) * hx;
      const double e1zb = E.v[1] + (U1yf.v[i0 - 1] - U1c.v[i0 - 1]) * hy;
      const double e2zb = E.v[2] + (U2c.v[i0] - U2c.v[i0 - 1]) * hz;
      const double e5xf = E.v[5] + 0.5 * ((U1xf.v[i0] - U1c.v[i0]) * hx + (U0xf.v[i0] - U0xfyb.v[i0]) * hy);
      const double e4xf = E.v[4] + 0.5 * ((U2xf.v[i0] - U2c.v[i0]) * hx + (U0xf.v[i0] - U0xf.v[i0 - 1]) * hz);
      const double e5yf = E.v[5] + 0.5 * ((U1yf.v[i0] - U1xbyf.v[i0]) * hx + (U0yf.v[i0] - U0c.v[i0]) * hy);
      const double e3yf = E.v[3] + 0.5 * ((U2yf.v[i0] - U2c.v[i0]) * hy + (U1yf.v[i0] - U1yf.v[i0 - 1]) * hz);
      const double e4zf = E.v[4] + 0.5 * ((U2c.v[i0 + 1] - U2xb.v[i0 + 1]) * hx + (U0c.v[i0 + 1] - U0c.v[i0]) * hz);
      const double e3zf = E.v[3] + 0.5 * ((U2c.v[i0 + 1] - U2yb.v[i0 + 1]) * hy + (U1c.v[i0 + 1] - U1c.v[i0]) * hz);
      // tau = (A + beta) eps + (B + gamma) tr(eps) I with the per-voxel effective moduli
      const double ac = Ac.v[i0] + beta, bc = Bc.v[i0] + gamma;
      const double trc = e0 + e1 + e2;
      const double t0 = e0 * ac + bc * trc, t1 = e1 * ac + bc * trc, t2 = e2 * ac + bc * trc;
      const double t3 = e3 * ac, t4 = e4 * ac, t5 = e5 * ac;
      const double t0xb = e0xb * (Axb.v[i0] + beta) + (Bxb.v[i0] + gamma) * (e0xb + e1xb + e2xb);
      const double t1yb = e1yb * (Ayb.v[i0] + beta) + (Byb.v[i0] + gamma) * (e0yb + e1yb + e2yb);
      const double t2zb = e2zb * (Ac.v[i0 - 1] + beta) + (Bc.v[i0 - 1] + gamma) * (e0zb + e1zb + e2zb);
      const double axf = Axf.v[i0] + beta, ayf = Ayf.v[i0] + beta, azf = Ac.v[i0 + 1] + beta;
      const double t5xf = e5xf * axf, t4xf = e4xf * axf;
      const double t5yf = e5yf * ayf, t3yf = e3yf * ayf;
      const double t4zf = e4zf * azf, t3zf = e3zf * azf;
      fout[s][0] = (t0 - t0xb) * hx + (t5yf - t5) * hy + (t4zf - t4) * hz;
      fout[s][1] = (t5xf - t5) * hx + (t1 - t1yb) * hy + (t3zf - t3) * hz;
      fout[s][2] = (t4xf - t4) * hx + (t3yf - t3) * hy + (t2 - t2zb) * hz;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double ey = second ? eout[1][c] : 0.0;
      acc[c] += eout[0][c] * eout[0][c] + ey * ey;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) st2(fo.p[c], p.off, make_double2(fout[0][c], second ? fout[1][c] : 0.0));
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}


// A = sum_p phi_p 2 mu_p, B = sum_p phi_p lambda_p with the Voigt rule's threshold (F:12736)
__global__ __launch_bounds__(kBlock) void k_effective_moduli(long n2, PhaseTable pt, FieldPtrs<kMaxPhases> phi,
                                                             FieldPtrs<2> mod) {
  const double threshold = 10 * 2.220446049250313e-16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    double2 A = make_double2(0.0, 0.0), B = make_double2(0.0, 0.0);
    for (int p = 0; p < pt.n; ++p) {
      const double2 f = ld2(phi.p[p], 2 * i);
      if (f.x > threshold) { A.x += 2 * f.x * pt.mu[p]; B.x += f.x * pt.lambda[p]; }
      if (f.y > threshold) { A.y += 2 * f.y * pt.mu[p]; B.y += f.y * pt.lambda[p]; }
    }
    st2(mod.p[0], 2 * i, A);
    st2(mod.p[1], 2 * i, B);
  }
}

}  // namespace

void launch_effective_moduli(const Grid& g, const PhaseTable& pt, const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<2>& mod,
                             hipStream_t s) {
  const long n2 = g.n / 2;
  long nb = (n2 + kBlock - 1) / kBlock;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(k_effective_moduli, dim3((unsigned)nb), dim3(kBlock), 0, s, n2, pt, phi, mod);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_u_fast(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& u, const FieldPtrs<2>& mod,
                   const FieldPtrs<3>& f, const Vec6& E, double* partial, double* sumsq6, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  hipLaunchKernelGGL(k_u_fast, dim3(nb), dim3(kBlock), 0, s, g, -2 * mu_0, -lambda_0, u, mod, f, E, partial,
                     chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

}  // namespace fg
