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

// The sweep holds, per (x,y) row it touches, the vector [z = k-1, k, k+1, k+2]: the pair (k, k+1) is one
// 16-byte load; the outer two values are the neighbouring lanes' pair halves, fetched by a wave shift.
// Only the first / last lane of a wave and the lanes at the ends of a z row load them from memory (edge
// loads, grouped so that they are issued together).  All 64 lanes stay active up to the stores.
// ODD: nz is odd, the last pair of a row holds one voxel and z+1 wraps to 0.
template <bool ODD>
__global__ __launch_bounds__(kBlock) void k_u_fast(Grid g, double beta, double gamma, FieldPtrs<3> u, FieldPtrs<2> mod,
                                                   FieldPtrs<3> fo, Vec6 E, double* partial, Sweep ry) {
  __shared__ double smem[4 * 6];
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx_raw = (run.first + it * run.stride) * kBlock + threadIdx.x;
    const long pidx = pidx_raw < npairs ? pidx_raw : npairs - 1;  // clamped lanes discard their result
    const PairPos p = pair_pos_tiled(pidx, g, ry);
    const bool valid = pidx_raw < npairs && p.k < g.nz;
    const bool second = !ODD || p.k + 1 < g.nz;
    const int lane = threadIdx.x & 63;
    const bool prev_ok = lane > 0 && p.k > 0;
    const bool next_ok = lane < 63 && p.k + 2 < g.nz && pidx_raw + 1 < npairs;
    const long xf = (p.i + 1 == g.nx ? -(long)(g.nx - 1) : 1L) * g.nyzp;
    const long xb = (p.i == 0 ? (long)(g.nx - 1) : -1L) * g.nyzp;
    const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
    const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
    const long ro = p.off - p.k;
    const int k = p.k;
    const int kb = k == 0 ? g.nz - 1 : k - 1;
    const int kf2 = (k + 2 >= g.nz) ? k + 2 - g.nz : k + 2;
    const double* const u0 = u.p[0];
    const double* const u1 = u.p[1];
    const double* const u2 = u.p[2];
    const double* const mA = mod.p[0];
    const double* const mB = mod.p[1];

    // edge loads: z = k-1 of {U0c, U0xf, U1c, U1yf, U2c, Ac, Bc}, z = k+2 of {U0c, U1c, U2c, U2yb, U2xb, Ac, Bc}
    double em[7] = {0, 0, 0, 0, 0, 0, 0}, ep[7] = {0, 0, 0, 0, 0, 0, 0};
    if (!prev_ok) {
      const long o = ro + kb;
      em[0] = u0[o]; em[1] = u0[o + xf]; em[2] = u1[o]; em[3] = u1[o + yf]; em[4] = u2[o]; em[5] = mA[o]; em[6] = mB[o];
    }
    if (!next_ok) {
      const long o = ro + kf2;
      ep[0] = u0[o]; ep[1] = u1[o]; ep[2] = u2[o]; ep[3] = u2[o + yb]; ep[4] = u2[o + xb]; ep[5] = mA[o]; ep[6] = mB[o];
    }
    const long rk = ro + k;
#define FG_ROW(name, a, off)                      \
    Row4 name;                                      \
    {                                               \
      const double2 d_ = ld2(a, rk + (off));        \
      name.v[1] = d_.x;                             \
      name.v[2] = d_.y;                             \
      if (ODD && !second) name.v[2] = (a)[ro + (off)]; \
    }
    FG_ROW(U0c, u0, 0) FG_ROW(U0xf, u0, xf) FG_ROW(U0yb, u0, yb) FG_ROW(U0xb, u0, xb) FG_ROW(U0xfyb, u0, xf + yb)
    FG_ROW(U0yf, u0, yf)
    FG_ROW(U1c, u1, 0) FG_ROW(U1yf, u1, yf) FG_ROW(U1xb, u1, xb) FG_ROW(U1xbyf, u1, xb + yf) FG_ROW(U1yb, u1, yb)
    FG_ROW(U1xf, u1, xf)
    FG_ROW(U2c, u2, 0) FG_ROW(U2yb, u2, yb) FG_ROW(U2xb, u2, xb) FG_ROW(U2xf, u2, xf) FG_ROW(U2yf, u2, yf)
    // effective moduli rows: A = sum_p phi_p 2 mu_p, B = sum_p phi_p lambda_p (precomputed per voxel)
    FG_ROW(Ac, mA, 0) FG_ROW(Bc, mB, 0) FG_ROW(Axb, mA, xb) FG_ROW(Bxb, mB, xb) FG_ROW(Axf, mA, xf)
    FG_ROW(Ayb, mA, yb) FG_ROW(Byb, mB, yb) FG_ROW(Ayf, mA, yf)
#undef FG_ROW
#define FG_PREV(row, i) { const double t_ = dpp_move<0x138>(row.v[2]); row.v[0] = prev_ok ? t_ : em[i]; }
#define FG_NEXT(row, i) { const double t_ = dpp_move<0x130>(row.v[1]); row.v[3] = next_ok ? t_ : ep[i]; }
    FG_PREV(U0c, 0) FG_PREV(U0xf, 1) FG_PREV(U1c, 2) FG_PREV(U1yf, 3) FG_PREV(U2c, 4) FG_PREV(Ac, 5) FG_PREV(Bc, 6)
    FG_NEXT(U0c, 0) FG_NEXT(U1c, 1) FG_NEXT(U2c, 2) FG_NEXT(U2yb, 3) FG_NEXT(U2xb, 4) FG_NEXT(Ac, 5) FG_NEXT(Bc, 6)
#undef FG_PREV
#undef FG_NEXT

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
    if (valid) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double ey = second ? eout[1][c] : 0.0;
        acc[c] += eout[0][c] * eout[0][c] + ey * ey;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) st2(fo.p[c], p.off, make_double2(fout[0][c], second ? fout[1][c] : 0.0));
    }
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
  if (g.nz % 2)
    hipLaunchKernelGGL(k_u_fast<true>, dim3(nb), dim3(kBlock), 0, s, g, -2 * mu_0, -lambda_0, u, mod, f, E, partial,
                       chunk_rows(g));
  else
    hipLaunchKernelGGL(k_u_fast<false>, dim3(nb), dim3(kBlock), 0, s, g, -2 * mu_0, -lambda_0, u, mod, f, E, partial,
                       chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

}  // namespace fg
