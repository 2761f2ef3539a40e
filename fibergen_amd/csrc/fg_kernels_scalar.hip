// Kernels of the scalar modes (mode = heat / porous, SURVEY 8f row 3, BASELINE config 5): the gradient
// field g = E + grad T has 3 components, the potential T one.  Voigt mixing of
// ScalarLinearIsotropicMaterialLaw (F:11158-11215: flux = mu * gradient).  Operation order of the reference
// throughout (this translation unit is compiled without FMA contraction).
//
// The loop state is the potential: one sweep T_k -> { sums of squares of g_k, f_{k+1} = div((C - C0) g_k) },
// then r2c / c2c / c2c, the scalar Green operator, c2c / c2c / c2r on ONE component give T_{k+1}.
#include "fg_kernels.h"

#include "fg_hip_util.h"
#include "fg_kernels_common.h"

namespace fg {

namespace {

constexpr double kVoigtThreshold = 10 * 2.220446049250313e-16;  // VoigtMixedMaterialLaw::init  F:12736

// VoigtMixedMaterialLaw<.,.,3>::PK1  F:12752-12761 over ScalarLinearIsotropicMaterialLaw::PK1  F:11182-11198
// for one component:  S (+)= E * ((phi * alpha) * mu), then  + beta * E  (calcStress  F:18160-18164).
__device__ __forceinline__ double sc_flux(const ScalarParams& sp, const double* ph, double g) {
  double P = 0.0;
  bool first = true;
  for (int p = 0; p < sp.n; ++p) {
    if (ph[p] <= kVoigtThreshold) continue;
    const double alpha_mu = (ph[p] * sp.alpha) * sp.mu[p];
    if (first) P = g * alpha_mu;
    else P += g * alpha_mu;
    first = false;
  }
  if (sp.beta != 0) P += sp.beta * g;
  return P;
}

struct PhiPair {
  double v[kMaxPhases][2];
};

// T_k -> sums of squares of g_k = E + grad+ T_k (epsOperatorStaggeredHeat  F:18697-18760, component_norm
// F:10127) and f = div-( (C - C0) g_k )  (calcStress  F:18134, divOperatorStaggeredHeat  F:18914-18975).
__global__ __launch_bounds__(kBlock) void k_sc_sweep(Grid g, ScalarParams sp, const double* T, FieldPtrs<kMaxPhases> phi,
                                                     double* f, Vec6 E, double* partial, Sweep sw) {
  __shared__ double smem[4 * 6];
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx = (run.first + it * run.stride) * kBlock + threadIdx.x;
    if (pidx >= npairs) continue;
    const PairPos p = pair_pos_tiled(pidx, g, sw);
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
    const Row4 Tc = load_row(T, ro, k, kb, kf2, second, true, true);
    const Row4 Txf = load_row(T, ro + xf, k, kb, kf2, second, false, false);
    const Row4 Txb = load_row(T, ro + xb, k, kb, kf2, second, false, false);
    const Row4 Tyf = load_row(T, ro + yf, k, kb, kf2, second, false, false);
    const Row4 Tyb = load_row(T, ro + yb, k, kb, kf2, second, false, false);
    double fo[2] = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 1 && !second) break;
      const int i0 = s + 1;
      double pc[kMaxPhases], pxb[kMaxPhases], pyb[kMaxPhases], pzb[kMaxPhases];
      const int kz = k + s;
      const int kzb = kz == 0 ? g.nz - 1 : kz - 1;
      for (int q = 0; q < sp.n; ++q) {
        pc[q] = phi.p[q][ro + kz];
        pxb[q] = phi.p[q][ro + xb + kz];
        pyb[q] = phi.p[q][ro + yb + kz];
        pzb[q] = phi.p[q][ro + kzb];
      }
      const double g0 = E.v[0] + (Txf.v[i0] - Tc.v[i0]) * hx;
      const double g1 = E.v[1] + (Tyf.v[i0] - Tc.v[i0]) * hy;
      const double g2 = E.v[2] + (Tc.v[i0 + 1] - Tc.v[i0]) * hz;
      acc[0] += g0 * g0;
      acc[1] += g1 * g1;
      acc[2] += g2 * g2;
      const double g0b = E.v[0] + (Tc.v[i0] - Txb.v[i0]) * hx;      // g_x at (i-1, j, k)
      const double g1b = E.v[1] + (Tc.v[i0] - Tyb.v[i0]) * hy;      // g_y at (i, j-1, k)
      const double g2b = E.v[2] + (Tc.v[i0] - Tc.v[i0 - 1]) * hz;   // g_z at (i, j, k-1)
      double y = (sc_flux(sp, pc, g0) - sc_flux(sp, pxb, g0b)) * hx;
      y += (sc_flux(sp, pc, g1) - sc_flux(sp, pyb, g1b)) * hy;
      y += (sc_flux(sp, pc, g2) - sc_flux(sp, pzb, g2b)) * hz;
      fo[s] = y;
    }
    st2(f, p.off, make_double2(fo[0], fo[1]));
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}

// T -> g = E + grad+ T (3 components) and the sums of squares
__global__ __launch_bounds__(kBlock) void k_sc_grad(Grid g, const double* T, FieldPtrs<3> out, Vec6 E, double* partial,
                                                    Sweep sw) {
  __shared__ double smem[4 * 6];
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx = (run.first + it * run.stride) * kBlock + threadIdx.x;
    if (pidx >= npairs) continue;
    const PairPos p = pair_pos_tiled(pidx, g, sw);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    // x neighbour through Grid::xw_hi: periodic in a whole grid, the spare plane of the right neighbour in an x-slab
    const long xf = (p.i + 1 == g.nx ? (long)(g.nx - g.xw_hi) - p.i : 1L) * g.nyzp;
    const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
    const long ro = p.off - p.k;
    const int k = p.k;
    const int kf2 = (k + 2 >= g.nz) ? k + 2 - g.nz : k + 2;
    const Row4 Tc = load_row(T, ro, k, 0, kf2, second, false, true);
    const Row4 Txf = load_row(T, ro + xf, k, 0, kf2, second, false, false);
    const Row4 Tyf = load_row(T, ro + yf, k, 0, kf2, second, false, false);
    double o[3][2] = {{0, 0}, {0, 0}, {0, 0}};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 1 && !second) break;
      const int i0 = s + 1;
      o[0][s] = E.v[0] + (Txf.v[i0] - Tc.v[i0]) * hx;
      o[1][s] = E.v[1] + (Tyf.v[i0] - Tc.v[i0]) * hy;
      o[2][s] = E.v[2] + (Tc.v[i0 + 1] - Tc.v[i0]) * hz;
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += o[c][s] * o[c][s];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) st2(out.p[c], p.off, make_double2(o[c][0], o[c][1]));
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}

// g -> flux P(g) * alpha + beta g (3 components), stored or (REDUCE: meanPK1  F:12312-12351) summed
template <bool REDUCE>
__global__ __launch_bounds__(kBlock) void k_sc_flux(Grid g, ScalarParams sp, FieldPtrs<3> gr, FieldPtrs<kMaxPhases> phi,
                                                    FieldPtrs<3> out, double* partial) {
  __shared__ double smem[4 * 6];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    if (p.k >= g.nz) continue;
    double o[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    for (int s = 0; s < 2; ++s) {
      if (p.k + s >= g.nz) break;
      double ph[kMaxPhases];
      for (int q = 0; q < sp.n; ++q) ph[q] = phi.p[q][p.off + s];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o[c][s] = sc_flux(sp, ph, gr.p[c][p.off + s]);
        acc[c] += o[c][s];
      }
    }
    if (!REDUCE) {
#pragma unroll
      for (int c = 0; c < 3; ++c) st2(out.p[c], p.off, make_double2(o[c][0], o[c][1]));
    }
  }
  if (REDUCE) {
    block_reduce<6>(acc, smem, OpSum());
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
    }
  }
}

// y = scale * div-(x)  (divOperatorStaggeredHeat  F:18914-18975 on scale * x)
__global__ __launch_bounds__(kBlock) void k_sc_div(Grid g, FieldPtrs<3> x, double scale, double* y) {
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    if (p.k >= g.nz) continue;
    const long xb = (p.i == 0 ? (long)(g.nx - 1) : -1L) * g.nyzp;
    const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
    const long ro = p.off - p.k;
    double o[2] = {0, 0};
    for (int s = 0; s < 2; ++s) {
      const int kz = p.k + s;
      if (kz >= g.nz) break;
      const int kzb = kz == 0 ? g.nz - 1 : kz - 1;
      double v = (scale * x.p[0][ro + kz] - scale * x.p[0][ro + xb + kz]) * hx;
      v += (scale * x.p[1][ro + kz] - scale * x.p[1][ro + yb + kz]) * hy;
      v += (scale * x.p[2][ro + kz] - scale * x.p[2][ro + kzb]) * hz;
      o[s] = v;
    }
    st2(y, p.off, make_double2(o[0], o[1]));
  }
}

// G0OperatorFourierStaggeredGeneralHeat  F:19779-19823: T_hat = c10 / |k|^2 f_hat, zero mode 0
// jj0: the slab driver's y-slab [nx][ny/P][nzc] holds the ky rows jj0 .. jj0 + g.ny - 1
__global__ __launch_bounds__(kBlock) void k_g0_heat(Grid g, cplx* fh, G0Tables tb, double c10, int jj0) {
  const long nfreq = (long)g.nx * g.ny * g.nzc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < nfreq; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / g.nzc;
    const int kk = (int)(idx - row * g.nzc);
    if (kk >= g.nzf) continue;
    const int ii = (int)(row / g.ny);
    const int jj = jj0 + (int)(row - (long)ii * g.ny);
    cplx e;
    if (ii == 0 && jj == 0 && kk == 0) {
      e = cmake(0.0, 0.0);
    } else {
      const double kpm0 = tb.kpm[0][ii], kpm1 = tb.kpm[1][jj], kpm2 = tb.kpm[2][kk];
      const double norm_kp2 = kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2;
      const double c1 = c10 / norm_kp2;
      e = cscale(c1, fh[idx]);
    }
    fh[idx] = e;
  }
}

// min / max over the voxels of the tangent eigenvalue sum_p phi_p mu_p (getRefMaterial  F:12153-12236 with the
// 3x3 tangent of the Voigt-mixed scalar law, a multiple of the identity)
__global__ __launch_bounds__(kBlock) void k_sc_minmax(Grid g, ScalarParams sp, FieldPtrs<kMaxPhases> phi, double* partial) {
  __shared__ double smem[4 * 2];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[2] = {1.0 / 0.0, 1.0 / 0.0};  // (min, -max)
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    for (int s = 0; s < 2; ++s) {
      if (p.k + s >= g.nz) continue;
      double ph[kMaxPhases];
      for (int q = 0; q < sp.n; ++q) ph[q] = phi.p[q][p.off + s];
      const double t = sc_flux(sp, ph, 1.0);  // dPK1 with W = 1: (phi * 1) * mu accumulated
      acc[0] = acc[0] < t ? acc[0] : t;
      acc[1] = acc[1] < -t ? acc[1] : -t;
    }
  }
  block_reduce<2>(acc, smem, OpMin());
  if (threadIdx.x == 0) {
    partial[(long)blockIdx.x * 2 + 0] = acc[0];
    partial[(long)blockIdx.x * 2 + 1] = acc[1];
  }
}

// CG in potential space (see k_cgu_dot in fg_kernels.hip): gradients are forward differences of the potentials.
//   MODE 0:  out[0] = sum grad a . (grad a - grad b)                              (p . (p - w))
//   MODE 1:  out[0..2] = sum (E + grad a)_c^2 ,  out[6] = sum grad b . grad b      (norms of g, r . r)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_sc_cg_dot(Grid g, const double* a, const double* b, Vec6 E, double* partial, Sweep sw) {
  __shared__ double smem[4 * 7];
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx = (run.first + it * run.stride) * kBlock + threadIdx.x;
    if (pidx >= npairs) continue;
    const PairPos p = pair_pos_tiled(pidx, g, sw);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    const long xf = (p.i + 1 == g.nx ? (long)(g.nx - g.xw_hi) - p.i : 1L) * g.nyzp;   // Grid::xw_hi: x-slabs read their spare plane
    const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
    const long ro = p.off - p.k;
    const int k = p.k;
    const int kf2 = (k + 2 >= g.nz) ? k + 2 - g.nz : k + 2;
    const Row4 Ac = load_row(a, ro, k, 0, kf2, second, false, true), Axf = load_row(a, ro + xf, k, 0, kf2, second, false, false);
    const Row4 Ayf = load_row(a, ro + yf, k, 0, kf2, second, false, false);
    const Row4 Bc = load_row(b, ro, k, 0, kf2, second, false, true), Bxf = load_row(b, ro + xf, k, 0, kf2, second, false, false);
    const Row4 Byf = load_row(b, ro + yf, k, 0, kf2, second, false, false);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 1 && !second) break;
      const int i0 = s + 1;
      const double ga[3] = {(Axf.v[i0] - Ac.v[i0]) * hx, (Ayf.v[i0] - Ac.v[i0]) * hy, (Ac.v[i0 + 1] - Ac.v[i0]) * hz};
      const double gb[3] = {(Bxf.v[i0] - Bc.v[i0]) * hx, (Byf.v[i0] - Bc.v[i0]) * hy, (Bc.v[i0 + 1] - Bc.v[i0]) * hz};
      if (MODE == 0) {
        acc[0] += ga[0] * (ga[0] - gb[0]) + ga[1] * (ga[1] - gb[1]) + ga[2] * (ga[2] - gb[2]);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double e = E.v[c] + ga[c];
          acc[c] += e * e;
        }
        acc[6] += gb[0] * gb[0] + gb[1] * gb[1] + gb[2] * gb[2];
      }
    }
  }
  block_reduce<7>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 7; ++c) partial[(long)blockIdx.x * 7 + c] = acc[c];
  }
}

// MODE 0:  x += a y ; r -= a (y - w)      MODE 1:  y = r + a y       (one-component fields)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_sc_cg_axpy(long n2, double* x, double* y, double* r, const double* w, double a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    if (MODE == 0) {
      double2 xv = ld2(x, 2 * i), rv = ld2(r, 2 * i);
      const double2 yv = ld2(y, 2 * i), wv = ld2(w, 2 * i);
      xv.x = xv.x + a * yv.x;
      xv.y = xv.y + a * yv.y;
      rv.x = rv.x - a * (yv.x - wv.x);
      rv.y = rv.y - a * (yv.y - wv.y);
      st2(x, 2 * i, xv);
      st2(r, 2 * i, rv);
    } else {
      double2 yv = ld2(y, 2 * i);
      const double2 rv = ld2(r, 2 * i);
      yv.x = rv.x + a * yv.x;
      yv.y = rv.y + a * yv.y;
      st2(y, 2 * i, yv);
    }
  }
}

int grid_cap(long nwork, int max_blocks) {
  long b = (nwork + kBlock - 1) / kBlock;
  if (b > max_blocks) b = max_blocks;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

void launch_sc_sweep(const Grid& g, const ScalarParams& sp, const double* T, const FieldPtrs<kMaxPhases>& phi, double* f,
                     const Vec6& E, double* partial, double* sumsq6, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  hipLaunchKernelGGL(k_sc_sweep, dim3(nb), dim3(kBlock), 0, s, g, sp, T, phi, f, E, partial, chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sc_grad(const Grid& g, const double* T, const FieldPtrs<3>& out, const Vec6& E, double* partial,
                    double* sumsq6, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  hipLaunchKernelGGL(k_sc_grad, dim3(nb), dim3(kBlock), 0, s, g, T, out, E, partial, chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sc_flux(const Grid& g, const ScalarParams& sp, const FieldPtrs<3>& gr, const FieldPtrs<kMaxPhases>& phi,
                    const FieldPtrs<3>& out, hipStream_t s) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  hipLaunchKernelGGL(k_sc_flux<false>, dim3(grid_cap(npairs, 1 << 20)), dim3(kBlock), 0, s, g, sp, gr, phi, out, nullptr);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sc_flux_mean(const Grid& g, const ScalarParams& sp, const FieldPtrs<3>& gr, const FieldPtrs<kMaxPhases>& phi,
                         double* partial, double* out6, hipStream_t s) {
  const int nb = reduce_blocks(g);
  hipLaunchKernelGGL(k_sc_flux<true>, dim3(nb), dim3(kBlock), 0, s, g, sp, gr, phi, gr, partial);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, out6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sc_div(const Grid& g, const FieldPtrs<3>& x, double scale, double* y, hipStream_t s) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  hipLaunchKernelGGL(k_sc_div, dim3(grid_cap(npairs, 1 << 20)), dim3(kBlock), 0, s, g, x, scale, y);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_g0_heat(const Grid& g, double* fh, const G0Tables& tb, double c10, hipStream_t s, int jj0) {
  const long nfreq = (long)g.nx * g.ny * g.nzc;
  hipLaunchKernelGGL(k_g0_heat, dim3(grid_cap(nfreq, 1 << 20)), dim3(kBlock), 0, s, g, reinterpret_cast<cplx*>(fh), tb, c10, jj0);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sc_cg_dot(int mode, const Grid& g, const double* a, const double* b, const Vec6& E, double* partial, double* out7,
                      hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  const Sweep sw = chunk_rows(g);
  if (mode == 0) hipLaunchKernelGGL((k_sc_cg_dot<0>), dim3(nb), dim3(kBlock), 0, s, g, a, b, E, partial, sw);
  else hipLaunchKernelGGL((k_sc_cg_dot<1>), dim3(nb), dim3(kBlock), 0, s, g, a, b, E, partial, sw);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 7, out7, s);
  FG_HIP_CHECK(hipGetLastError());
}

// the same updates out of place on `count` doubles from offset `off`, with the coefficient formed on the device (x-slabs with
// the fused CG sweeps: the spare planes of the alternate buffers, see k_cgu_axpy_oop)
//   MODE 0:  xo = x + a y ;  ro = r - a (y - w)          MODE 1:  xo = r + a y
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_sc_cg_axpy_oop(long n2, long off, const double* x, const double* y, const double* r,
                                                           const double* w, double* xo, double* ro, const double* sc, int i_num,
                                                           int i_den, double nvox, double small) {
  const double a = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    const long o = off + 2 * i;
    const double2 yv = ld2(y, o), rv = ld2(r, o);
    if (MODE == 0) {
      const double2 xv = ld2(x, o), wv = ld2(w, o);
      st2(xo, o, make_double2(xv.x + a * yv.x, xv.y + a * yv.y));
      st2(ro, o, make_double2(rv.x - a * (yv.x - wv.x), rv.y - a * (yv.y - wv.y)));
    } else {
      st2(xo, o, make_double2(rv.x + a * yv.x, rv.y + a * yv.y));
    }
  }
}

void launch_sc_cg_axpy_oop(int mode, const double* x, const double* y, const double* r, const double* w, double* xo, double* ro,
                           const double* sc, int i_num, int i_den, double nvox, double small, long off, long count, hipStream_t s) {
  const long n2 = count / 2;
  if (n2 <= 0) return;
  const int nb = grid_cap(n2, 1 << 14);
  if (mode == 0) hipLaunchKernelGGL((k_sc_cg_axpy_oop<0>), dim3(nb), dim3(kBlock), 0, s, n2, off, x, y, r, w, xo, ro, sc, i_num, i_den, nvox, small);
  else hipLaunchKernelGGL((k_sc_cg_axpy_oop<1>), dim3(nb), dim3(kBlock), 0, s, n2, off, x, y, r, w, xo, ro, sc, i_num, i_den, nvox, small);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sc_cg_axpy(int mode, const Grid& g, double* x, double* y, double* r, const double* w, double a, hipStream_t s,
                       long count) {
  const long n2 = (count > 0 ? count : g.n) / 2;
  const dim3 grid(grid_cap(n2, 1 << 16));
  if (mode == 0) hipLaunchKernelGGL((k_sc_cg_axpy<0>), grid, dim3(kBlock), 0, s, n2, x, y, r, w, a);
  else hipLaunchKernelGGL((k_sc_cg_axpy<1>), grid, dim3(kBlock), 0, s, n2, x, y, r, w, a);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sc_minmax(const Grid& g, const ScalarParams& sp, const FieldPtrs<kMaxPhases>& phi, double* partial,
                      double* out2, hipStream_t s) {
  const int nb = reduce_blocks(g);
  hipLaunchKernelGGL(k_sc_minmax, dim3(nb), dim3(kBlock), 0, s, g, sp, phi, partial);
  FG_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_fold<OpMin>, dim3(1), dim3(kBlock), 0, s, partial, nb, 2, 1.0 / 0.0, out2);
  FG_HIP_CHECK(hipGetLastError());
}

}  // namespace fg
