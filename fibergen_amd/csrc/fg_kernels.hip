// Streaming kernels of the Lippmann-Schwinger iteration for gfx950.
//
// All of them are HBM-bound float64 sweeps over SoA components laid out like the
// reference's TensorField (F:9584-10282): [nx][ny][nzp], z fastest, nzp = 2*(nz/2+1).
// Threads own one double2 (two z-neighbouring voxels) => 16-byte coalesced accesses;
// the z padding pair is skipped.  Reductions are two-stage with a fixed tree, so
// results are bitwise reproducible from run to run.
#include "fg_kernels.h"

#include <algorithm>

#include "fg_hip_util.h"
#include "fg_kernels_common.h"

namespace fg {

namespace {
#ifndef FG_K1_WAVES
#define FG_K1_WAVES 2
#endif

// ----------------------------------------------------------------------------- stress
// calcStress  F:18134-18184.  REDUCE = false: tau <- P(eps) - C0:eps written out.
// REDUCE = 1: per-block partial sums of P (meanPK1  F:12312-12351, alpha already /N).
// REDUCE = 2: per-block partial sums of the energy density 1/2 P:eps (meanW  F:12239-12262 with the mixing rule's W: for
// Voigt mixing sum_p phi_p 1/2 (C_p eps):eps = 1/2 P:eps; for laminate mixing c1 W1(F1) + c2 W2(F2) = 1/2 P:eps as well, the
// cross term c1 c2 a.(sigma_1 - sigma_2) n vanishing with the traction jump the interface solve removes), in component 0.
// MIX / NPH are compile-time so the Voigt two-phase sweep keeps a small register footprint.
template <int REDUCE, int MIX, int NPH>
__global__ __launch_bounds__(kBlock) void k_stress(Grid g, StressParams sp, FieldPtrs<6> eps, FieldPtrs<kMaxPhases> phi,
                                                   FieldPtrs<3> normals, FieldPtrs<6> tau, double* partial,
                                                   int* error_flag) {
  __shared__ double smem[4 * 6];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    if (p.k >= g.nz) continue;  // padding pair
    const bool second = p.k + 1 < g.nz;
    double2 e[6], f[NPH], nn[3];
#pragma unroll
    for (int c = 0; c < 6; ++c) e[c] = ld2(eps.p[c], p.off);
#pragma unroll
    for (int q = 0; q < NPH; ++q) f[q] = q < sp.pt.n ? ld2(phi.p[q], p.off) : make_double2(0.0, 0.0);
    nn[0] = nn[1] = nn[2] = make_double2(0.0, 0.0);
    if (MIX == kMixLaminate) {
      // the normal is only used at composite voxels (some 0 < phi < 1): skip the 24 B/voxel elsewhere
      bool mixed = false;
#pragma unroll
      for (int q = 0; q < NPH; ++q)
        mixed = mixed || (f[q].x != 0.0 && f[q].x != 1.0) || (f[q].y != 0.0 && f[q].y != 1.0);
      if (mixed) {
#pragma unroll
        for (int c = 0; c < 3; ++c) nn[c] = ld2(normals.p[c], p.off);
      }
    }
    double F[6], ph[NPH], nv[3], P0[6], P1[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) F[c] = e[c].x;
#pragma unroll
    for (int q = 0; q < NPH; ++q) ph[q] = f[q].x;
    nv[0] = nn[0].x; nv[1] = nn[1].x; nv[2] = nn[2].x;
    int err = stress_voxel<NPH>(F, ph, nv, sp, P0);
    double w = 0.0;
    if (REDUCE == 2) w = 0.5 * (P0[0] * F[0] + P0[1] * F[1] + P0[2] * F[2] + 2 * (P0[3] * F[3] + P0[4] * F[4] + P0[5] * F[5]));
    if (second) {
#pragma unroll
      for (int c = 0; c < 6; ++c) F[c] = e[c].y;
#pragma unroll
      for (int q = 0; q < NPH; ++q) ph[q] = f[q].y;
      nv[0] = nn[0].y; nv[1] = nn[1].y; nv[2] = nn[2].y;
      err |= stress_voxel<NPH>(F, ph, nv, sp, P1);
      if (REDUCE == 2) w += 0.5 * (P1[0] * F[0] + P1[1] * F[1] + P1[2] * F[2] + 2 * (P1[3] * F[3] + P1[4] * F[4] + P1[5] * F[5]));
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) P1[c] = 0.0;
    }
    if (err) atomicOr(error_flag, 1);
    if (REDUCE == 2) {
      acc[0] += w;
    } else if (REDUCE) {
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[c] += P0[c] + P1[c];
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) st2(tau.p[c], p.off, make_double2(P0[c], P1[c]));
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

// ----------------------------------------------------------------------------- divergence
// divOperatorStaggered  F:18853-18908:
//  f0 = D-x t0 + D+y t5 + D+z t4 ; f1 = D+x t5 + D-y t1 + D+z t3 ; f2 = D+x t4 + D+y t3 + D-z t2
__global__ __launch_bounds__(kBlock) void k_div(Grid g, FieldPtrs<6> t, FieldPtrs<3> f, XHalo h, Sweep ry) {
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx = (run.first + it * run.stride) * kBlock + threadIdx.x;
    if (pidx >= npairs) continue;
    const PairPos p = pair_pos_tiled(pidx, g, ry);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    // periodic neighbour offsets (F:14867-14891)
    const long xf = (p.i + 1 == g.nx ? -(long)(g.nx - 1) : 1L) * g.nyzp;
    const long xb = (p.i == 0 ? (long)(g.nx - 1) : -1L) * g.nyzp;
    const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
    const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
    const long rowoff = p.off - p.k;
    const int kb = p.k == 0 ? g.nz - 1 : p.k - 1;               // z-1 of the first voxel
    const int kf2 = (p.k + 2 >= g.nz) ? p.k + 2 - g.nz : p.k + 2;  // z+1 of the second voxel

    const double2 t0 = ld2(t.p[0], p.off), t1 = ld2(t.p[1], p.off), t2 = ld2(t.p[2], p.off);
    const double2 t3 = ld2(t.p[3], p.off), t4 = ld2(t.p[4], p.off), t5 = ld2(t.p[5], p.off);
    // x neighbours of the first / last local plane come from the halo planes when the grid is an x-slab
    const long inplane = p.off - (long)p.i * g.nyzp;
    const bool lo = p.i == 0 && h.lo[0] != nullptr, hi = p.i + 1 == g.nx && h.hi[0] != nullptr;
    const double2 t0xb = lo ? ld2(h.lo[0], inplane) : ld2(t.p[0], p.off + xb);
    const double2 t5yf = ld2(t.p[5], p.off + yf);
    const double2 t5xf = hi ? ld2(h.hi[0], inplane) : ld2(t.p[5], p.off + xf);
    const double2 t1yb = ld2(t.p[1], p.off + yb);
    const double2 t4xf = hi ? ld2(h.hi[1], inplane) : ld2(t.p[4], p.off + xf);
    const double2 t3yf = ld2(t.p[3], p.off + yf);
    const double t4zf2 = t.p[4][rowoff + kf2];
    const double t3zf2 = t.p[3][rowoff + kf2];
    const double t2zb = t.p[2][rowoff + kb];
    // z+1 of the first voxel is the second voxel, or (odd nz, last pair) wraps to k = 0
    const double t4zf1 = second ? t4.y : t.p[4][rowoff];
    const double t3zf1 = second ? t3.y : t.p[3][rowoff];

    double2 f0, f1, f2;
    f0.x = (t0.x - t0xb.x) * hx + (t5yf.x - t5.x) * hy + (t4zf1 - t4.x) * hz;
    f1.x = (t5xf.x - t5.x) * hx + (t1.x - t1yb.x) * hy + (t3zf1 - t3.x) * hz;
    f2.x = (t4xf.x - t4.x) * hx + (t3yf.x - t3.x) * hy + (t2.x - t2zb) * hz;
    f0.y = (t0.y - t0xb.y) * hx + (t5yf.y - t5.y) * hy + (t4zf2 - t4.y) * hz;
    f1.y = (t5xf.y - t5.y) * hx + (t1.y - t1yb.y) * hy + (t3zf2 - t3.y) * hz;
    f2.y = (t4xf.y - t4.y) * hx + (t3yf.y - t3.y) * hy + (t2.y - t2.x) * hz;
    st2(f.p[0], p.off, f0);
    st2(f.p[1], p.off, f1);
    st2(f.p[2], p.off, f2);
  }
}

// ----------------------------------------------------------------------------- stress + divergence, fused
// f = div[(C - C0) : eps] in one sweep for Voigt mixing: the polarisation is never stored.  Each thread
// re-evaluates the few tau components it needs at the six neighbours (a Hooke law is ~10 flops; the
// neighbour loads are L1/L2 hits thanks to the L2-aware sweep), which removes the 6-component write and
// read of tau: 176 -> 88 algorithmic bytes per voxel.  Bit-identical to k_stress followed by k_div.
template <int NPH>
struct VoxelIn {
  double2 e[6];
  double2 f[NPH];
};

template <int NPH, bool SUM>
__global__ __launch_bounds__(kBlock) void k_stress_div_voigt(Grid g, StressParams sp, FieldPtrs<6> eps,
                                                             FieldPtrs<kMaxPhases> phi, FieldPtrs<3> fo, double* partial,
                                                             Sweep ry) {
  __shared__ double smem[SUM ? 4 * 6 : 1];
  double acc[6] = {0, 0, 0, 0, 0, 0};   // SUM: sums of the polarisation components (viscosity mode needs <tau>)
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
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
    const long rowoff = p.off - p.k;
    const int kb = p.k == 0 ? g.nz - 1 : p.k - 1;
    const int kf2 = (p.k + 2 >= g.nz) ? p.k + 2 - g.nz : p.k + 2;

    double ph[NPH];
    // ---- centre pair: all six components
    double2 e[6], fc[NPH];
#pragma unroll
    for (int c = 0; c < 6; ++c) e[c] = ld2(eps.p[c], p.off);
#pragma unroll
    for (int q = 0; q < NPH; ++q) fc[q] = q < sp.pt.n ? ld2(phi.p[q], p.off) : make_double2(0.0, 0.0);
    double2 t[6];
#pragma unroll
    for (int q = 0; q < NPH; ++q) ph[q] = fc[q].x;
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c].x = voigt_tau_normal<NPH>(e[c].x, e[0].x, e[1].x, e[2].x, ph, sp);
#pragma unroll
    for (int c = 3; c < 6; ++c) t[c].x = voigt_tau_shear<NPH>(e[c].x, ph, sp);
#pragma unroll
    for (int q = 0; q < NPH; ++q) ph[q] = fc[q].y;
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c].y = voigt_tau_normal<NPH>(e[c].y, e[0].y, e[1].y, e[2].y, ph, sp);
#pragma unroll
    for (int c = 3; c < 6; ++c) t[c].y = voigt_tau_shear<NPH>(e[c].y, ph, sp);

    // ---- neighbours: only the components the divergence uses
    auto normal_at = [&](long off, int c) {  // tau_c (c < 3) of the pair at element offset off
      const double2 a0 = ld2(eps.p[0], off), a1 = ld2(eps.p[1], off), a2 = ld2(eps.p[2], off);
      double2 pf[NPH];
#pragma unroll
      for (int q = 0; q < NPH; ++q) pf[q] = q < sp.pt.n ? ld2(phi.p[q], off) : make_double2(0.0, 0.0);
      double pp[NPH];
      double2 r;
      const double2 ac = c == 0 ? a0 : (c == 1 ? a1 : a2);
#pragma unroll
      for (int q = 0; q < NPH; ++q) pp[q] = pf[q].x;
      r.x = voigt_tau_normal<NPH>(ac.x, a0.x, a1.x, a2.x, pp, sp);
#pragma unroll
      for (int q = 0; q < NPH; ++q) pp[q] = pf[q].y;
      r.y = voigt_tau_normal<NPH>(ac.y, a0.y, a1.y, a2.y, pp, sp);
      return r;
    };
    auto shear2_at = [&](long off, int ca, int cb, double2* ra, double2* rb) {  // two shear components
      const double2 a = ld2(eps.p[ca], off), b = ld2(eps.p[cb], off);
      double2 pf[NPH];
#pragma unroll
      for (int q = 0; q < NPH; ++q) pf[q] = q < sp.pt.n ? ld2(phi.p[q], off) : make_double2(0.0, 0.0);
      double pp[NPH];
#pragma unroll
      for (int q = 0; q < NPH; ++q) pp[q] = pf[q].x;
      ra->x = voigt_tau_shear<NPH>(a.x, pp, sp);
      rb->x = voigt_tau_shear<NPH>(b.x, pp, sp);
#pragma unroll
      for (int q = 0; q < NPH; ++q) pp[q] = pf[q].y;
      ra->y = voigt_tau_shear<NPH>(a.y, pp, sp);
      rb->y = voigt_tau_shear<NPH>(b.y, pp, sp);
    };
    const double2 t0xb = normal_at(p.off + xb, 0);
    const double2 t1yb = normal_at(p.off + yb, 1);
    double2 t5xf, t4xf, t5yf, t3yf;
    shear2_at(p.off + xf, 5, 4, &t5xf, &t4xf);
    shear2_at(p.off + yf, 5, 3, &t5yf, &t3yf);
    // z neighbours inside the row: voxel k-1 (tau2) and voxel k+2 (tau4, tau3), scalar
    double t2zb, t4zf2, t3zf2;
    {
#pragma unroll
      for (int q = 0; q < NPH; ++q) ph[q] = q < sp.pt.n ? phi.p[q][rowoff + kb] : 0.0;
      const double a0 = eps.p[0][rowoff + kb], a1 = eps.p[1][rowoff + kb], a2 = eps.p[2][rowoff + kb];
      t2zb = voigt_tau_normal<NPH>(a2, a0, a1, a2, ph, sp);
#pragma unroll
      for (int q = 0; q < NPH; ++q) ph[q] = q < sp.pt.n ? phi.p[q][rowoff + kf2] : 0.0;
      t4zf2 = voigt_tau_shear<NPH>(eps.p[4][rowoff + kf2], ph, sp);
      t3zf2 = voigt_tau_shear<NPH>(eps.p[3][rowoff + kf2], ph, sp);
    }
    double t4zf1 = t[4].y, t3zf1 = t[3].y;
    if (!second) {  // odd nz, last pair: z+1 of the first voxel wraps to k = 0
#pragma unroll
      for (int q = 0; q < NPH; ++q) ph[q] = q < sp.pt.n ? phi.p[q][rowoff] : 0.0;
      t4zf1 = voigt_tau_shear<NPH>(eps.p[4][rowoff], ph, sp);
      t3zf1 = voigt_tau_shear<NPH>(eps.p[3][rowoff], ph, sp);
    }

    double2 f0, f1, f2;  // same expressions as k_div  (F:18853-18908)
    f0.x = (t[0].x - t0xb.x) * hx + (t5yf.x - t[5].x) * hy + (t4zf1 - t[4].x) * hz;
    f1.x = (t5xf.x - t[5].x) * hx + (t[1].x - t1yb.x) * hy + (t3zf1 - t[3].x) * hz;
    f2.x = (t4xf.x - t[4].x) * hx + (t3yf.x - t[3].x) * hy + (t[2].x - t2zb) * hz;
    f0.y = (t[0].y - t0xb.y) * hx + (t5yf.y - t[5].y) * hy + (t4zf2 - t[4].y) * hz;
    f1.y = (t5xf.y - t[5].y) * hx + (t[1].y - t1yb.y) * hy + (t3zf2 - t[3].y) * hz;
    f2.y = (t4xf.y - t[4].y) * hx + (t3yf.y - t[3].y) * hy + (t[2].y - t[2].x) * hz;
    if (!second) f0.y = f1.y = f2.y = 0.0;
    st2(fo.p[0], p.off, f0);
    st2(fo.p[1], p.off, f1);
    st2(fo.p[2], p.off, f2);
    if (SUM) {
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[c] += t[c].x + (second ? t[c].y : 0.0);
    }
  }
  if (SUM) {
    block_reduce<6>(acc, smem, OpSum());
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
    }
  }
}

// ----------------------------------------------------------------------------- displacement-based pass
// f = div[(C - C0) : (E + sym grad u)] straight from the displacement of the previous pass, plus the
// sums of squares of the strain.  The strain field is never stored inside the loop: the strain operator
// of pass k (epsOperatorStaggered F:18614-18692), the polarisation (calcStress F:18134-18184, Voigt
// mixing) and the divergence (divOperatorStaggered F:18853-18908) of pass k+1 become one sweep that
// reads 3 + n_phase arrays and writes 3 (64 B/voxel for two phases instead of 72 + 88).  Every value is
// computed with the expressions of k_eps_norm / stress_voxel / k_div, so the pass is bit-identical to
// the three-kernel form.  Row vectors hold the values at z = k-1, k, k+1, k+2 of one (x,y) row.
template <int NPH>
struct PhiRows {
  Row4 r[NPH];
};

template <int NPH>
__device__ __forceinline__ PhiRows<NPH> load_phi(const FieldPtrs<kMaxPhases>& phi, int n, long ro, int k, int kb, int kf2,
                                                 bool second, bool wide) {
  PhiRows<NPH> o;
#pragma unroll
  for (int q = 0; q < NPH; ++q) {
    if (q < n) o.r[q] = load_row(phi.p[q], ro, k, kb, kf2, second, wide, wide);
    else o.r[q].v[0] = o.r[q].v[1] = o.r[q].v[2] = o.r[q].v[3] = 0.0;
  }
  return o;
}

template <int NPH>
__global__ __launch_bounds__(kBlock, FG_K1_WAVES) void k_u_stress_div_voigt(Grid g, StressParams sp, FieldPtrs<3> u,
                                                               FieldPtrs<kMaxPhases> phi, FieldPtrs<3> fo, Vec6 E,
                                                               double* partial, Sweep ry) {
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
    const PhiRows<NPH> Pc = load_phi<NPH>(phi, sp.pt.n, ro, k, kb, kf2, second, true);
    const PhiRows<NPH> Pxb = load_phi<NPH>(phi, sp.pt.n, ro + xb, k, kb, kf2, second, false);
    const PhiRows<NPH> Pxf = load_phi<NPH>(phi, sp.pt.n, ro + xf, k, kb, kf2, second, false);
    const PhiRows<NPH> Pyb = load_phi<NPH>(phi, sp.pt.n, ro + yb, k, kb, kf2, second, false);
    const PhiRows<NPH> Pyf = load_phi<NPH>(phi, sp.pt.n, ro + yf, k, kb, kf2, second, false);

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
      double pc[NPH], pxb[NPH], pxf[NPH], pyb[NPH], pyf[NPH], pzb[NPH], pzf[NPH];
#pragma unroll
      for (int q = 0; q < NPH; ++q) {
        pc[q] = Pc.r[q].v[i0]; pzb[q] = Pc.r[q].v[i0 - 1]; pzf[q] = Pc.r[q].v[i0 + 1];
        pxb[q] = Pxb.r[q].v[i0]; pxf[q] = Pxf.r[q].v[i0]; pyb[q] = Pyb.r[q].v[i0]; pyf[q] = Pyf.r[q].v[i0];
      }
      const double t0 = voigt_tau_normal<NPH>(e0, e0, e1, e2, pc, sp);
      const double t1 = voigt_tau_normal<NPH>(e1, e0, e1, e2, pc, sp);
      const double t2 = voigt_tau_normal<NPH>(e2, e0, e1, e2, pc, sp);
      const double t3 = voigt_tau_shear<NPH>(e3, pc, sp), t4 = voigt_tau_shear<NPH>(e4, pc, sp);
      const double t5 = voigt_tau_shear<NPH>(e5, pc, sp);
      const double t0xb = voigt_tau_normal<NPH>(e0xb, e0xb, e1xb, e2xb, pxb, sp);
      const double t1yb = voigt_tau_normal<NPH>(e1yb, e0yb, e1yb, e2yb, pyb, sp);
      const double t2zb = voigt_tau_normal<NPH>(e2zb, e0zb, e1zb, e2zb, pzb, sp);
      const double t5xf = voigt_tau_shear<NPH>(e5xf, pxf, sp), t4xf = voigt_tau_shear<NPH>(e4xf, pxf, sp);
      const double t5yf = voigt_tau_shear<NPH>(e5yf, pyf, sp), t3yf = voigt_tau_shear<NPH>(e3yf, pyf, sp);
      const double t4zf = voigt_tau_shear<NPH>(e4zf, pzf, sp), t3zf = voigt_tau_shear<NPH>(e3zf, pzf, sp);
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

// ----------------------------------------------------------------------------- Green operator
// G0OperatorFourierStaggeredGeneral  F:19834-19927, in place on 3 complex components.
// Layout [nx][ny][nzc]; for the y-slab of the slab-decomposed transform g.ny is the slab thickness and the global
// ky is jj0 + local row (the zero mode lives on the slab with jj0 == 0).
__global__ __launch_bounds__(kBlock) void k_g0(Grid g, FieldPtrs<3> fh, G0Tables tb, double c10, double c20,
                                               G0Layout lay) {
  const long nfreq = lay.transposed ? (long)lay.nyl * g.nx * g.nzc : (long)g.nx * g.ny * g.nzc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < nfreq; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / g.nzc;
    const int kk = (int)(idx - row * g.nzc);
    if (kk >= g.nzf) continue;  // row padding
    int ii, jj;
    if (lay.transposed) {
      const int jl = (int)(row / g.nx);
      ii = (int)(row - (long)jl * g.nx);
      jj = lay.jj0 + jl;
    } else {
      ii = (int)(row / g.ny);
      jj = lay.jj0 + (int)(row - (long)ii * g.ny);   // jj0 != 0: y-slab [nx][ny/P][nzc] of the slab driver
    }
    cplx* c0 = reinterpret_cast<cplx*>(fh.p[0]);
    cplx* c1 = reinterpret_cast<cplx*>(fh.p[1]);
    cplx* c2 = reinterpret_cast<cplx*>(fh.p[2]);
    cplx e0, e1, e2;
    if (ii == 0 && jj == 0 && kk == 0) {
      e0 = e1 = e2 = cmake(0.0, 0.0);  // zero frequency  F:19924-19926
    } else {
      g0_point(c0[idx], c1[idx], c2[idx], tb.kpm[0][ii], tb.kpm[1][jj], tb.kpm[2][kk], tb.kp[0][ii], tb.kp[1][jj],
               tb.kp[2][kk], c10, c20, &e0, &e1, &e2);
    }
    c0[idx] = e0;
    c1[idx] = e1;
    c2[idx] = e2;
  }
}

// ----------------------------------------------------------------------------- collocated Gamma operator
// GammaOperatorFourierCollocated  F:19381-19608 (freq_hack off): per frequency the real symmetric 6x6
// Gamma0_hat built from xi = m/d (F:19434-19456), eta_hat_i = sum_j g_ij tau_hat_j (x2 for the shear columns)
// + beta tau_hat_i, in place on the six complex components; the zero frequency is set to E (F:19605-19607).
__global__ __launch_bounds__(kBlock) void k_gamma_collocated(Grid g, FieldPtrs<6> th, XiTables xt, double c10, double c20,
                                                             double beta, Vec6 E) {
  const long nfreq = (long)g.nx * g.ny * g.nzc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < nfreq; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / g.nzc;
    const int kk = (int)(idx - row * g.nzc);
    if (kk >= g.nzf) continue;  // row padding
    const int ii = (int)(row / g.ny);
    const int jj = (int)(row - (long)ii * g.ny);
    cplx t[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) t[c] = reinterpret_cast<const cplx*>(th.p[c])[idx];
    cplx ey[6];
    if (ii == 0 && jj == 0 && kk == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) ey[c] = cmake(E.v[c], 0.0);
    } else {
      const double xi0 = xt.xi[0][ii], xi1 = xt.xi[1][jj], xi2 = xt.xi[2][kk];
      const double xi00 = xi0 * xi0, xi01 = xi0 * xi1, xi11 = xi1 * xi1;
      const double xi02 = xi0 * xi2, xi12 = xi1 * xi2, xi22 = xi2 * xi2;
      const double norm_xi2 = xi00 + xi11 + xi22;
      const double c1 = c10 / (norm_xi2);
      const double c12 = c1 * 2;
      const double c2 = c20 / (norm_xi2 * norm_xi2);
      const double c3 = (c12 + c2 * xi00);
      const double c4 = (c12 + c2 * xi11);
      const double c5 = (c12 + c2 * xi22);
      double G[6][6];
      G[0][0] = (c12 + c3) * xi00;
      G[1][0] = c2 * xi00 * xi11;
      G[2][0] = c2 * xi00 * xi22;
      G[3][0] = c2 * xi00 * xi12;
      G[4][0] = c3 * xi02;
      G[5][0] = c3 * xi01;
      G[1][1] = (c12 + c4) * xi11;
      G[2][1] = c2 * xi11 * xi22;
      G[3][1] = c4 * xi12;
      G[4][1] = c2 * xi11 * xi02;
      G[5][1] = c4 * xi01;
      G[2][2] = (c12 + c5) * xi22;
      G[3][2] = c5 * xi12;
      G[4][2] = c5 * xi02;
      G[5][2] = c2 * xi22 * xi01;
      G[3][3] = (c1 * (xi11 + xi22) + c2 * xi11 * xi22);
      G[4][3] = (c1 + c2 * xi22) * xi01;
      G[5][3] = (c1 + c2 * xi11) * xi02;
      G[4][4] = (c1 * (xi00 + xi22) + c2 * xi00 * xi22);
      G[5][4] = (c1 + c2 * xi00) * xi12;
      G[5][5] = (c1 * (xi00 + xi11) + c2 * xi00 * xi11);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i + 1; j < 6; ++j) G[i][j] = G[j][i];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        // tau0*g0 + tau1*g1 + tau2*g2 + (tau3*g3 + tau4*g4 + tau5*g5)*2, real and imaginary parts alike
        const double ar = t[0].re * G[i][0] + t[1].re * G[i][1] + t[2].re * G[i][2] +
                          (t[3].re * G[i][3] + t[4].re * G[i][4] + t[5].re * G[i][5]) * 2.0;
        const double ai = t[0].im * G[i][0] + t[1].im * G[i][1] + t[2].im * G[i][2] +
                          (t[3].im * G[i][3] + t[4].im * G[i][4] + t[5].im * G[i][5]) * 2.0;
        ey[i] = cmake(ar + beta * t[i].re, ai + beta * t[i].im);
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) reinterpret_cast<cplx*>(th.p[c])[idx] = ey[c];
  }
}

// ----------------------------------------------------------------------------- strain + norm
// epsOperatorStaggered  F:18614-18692, followed by eps += R (applyBCProjector  F:20263-20270)
// and the per-component sums of squares of component_norm (F:10088-10138) fused in.
__global__ __launch_bounds__(kBlock) void k_eps_norm(Grid g, FieldPtrs<3> u, FieldPtrs<6> eps, Vec6 E, Vec6 R, int add_R,
                                                     double* partial, XHalo h, Sweep ry) {
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
    const long rowoff = p.off - p.k;
    const int kb = p.k == 0 ? g.nz - 1 : p.k - 1;
    const int kf2 = (p.k + 2 >= g.nz) ? p.k + 2 - g.nz : p.k + 2;

    const double2 u0 = ld2(u.p[0], p.off), u1 = ld2(u.p[1], p.off), u2 = ld2(u.p[2], p.off);
    const long inplane = p.off - (long)p.i * g.nyzp;
    const bool lo = p.i == 0 && h.lo[0] != nullptr, hi = p.i + 1 == g.nx && h.hi[0] != nullptr;
    const double2 u0xf = hi ? ld2(h.hi[0], inplane) : ld2(u.p[0], p.off + xf);
    const double2 u1xb = lo ? ld2(h.lo[0], inplane) : ld2(u.p[1], p.off + xb);
    const double2 u2xb = lo ? ld2(h.lo[1], inplane) : ld2(u.p[2], p.off + xb);
    const double2 u0yb = ld2(u.p[0], p.off + yb), u1yf = ld2(u.p[1], p.off + yf), u2yb = ld2(u.p[2], p.off + yb);
    const double u0zb = u.p[0][rowoff + kb], u1zb = u.p[1][rowoff + kb];
    const double u2zf2 = u.p[2][rowoff + kf2];
    const double u2zf1 = second ? u2.y : u.p[2][rowoff];

    double2 e[6];
    e[3].x = E.v[3] + 0.5 * ((u2.x - u2yb.x) * hy + (u1.x - u1zb) * hz);
    e[4].x = E.v[4] + 0.5 * ((u2.x - u2xb.x) * hx + (u0.x - u0zb) * hz);
    e[5].x = E.v[5] + 0.5 * ((u1.x - u1xb.x) * hx + (u0.x - u0yb.x) * hy);
    e[0].x = E.v[0] + (u0xf.x - u0.x) * hx;
    e[1].x = E.v[1] + (u1yf.x - u1.x) * hy;
    e[2].x = E.v[2] + (u2zf1 - u2.x) * hz;
    e[3].y = E.v[3] + 0.5 * ((u2.y - u2yb.y) * hy + (u1.y - u1.x) * hz);
    e[4].y = E.v[4] + 0.5 * ((u2.y - u2xb.y) * hx + (u0.y - u0.x) * hz);
    e[5].y = E.v[5] + 0.5 * ((u1.y - u1xb.y) * hx + (u0.y - u0yb.y) * hy);
    e[0].y = E.v[0] + (u0xf.y - u0.y) * hx;
    e[1].y = E.v[1] + (u1yf.y - u1.y) * hy;
    e[2].y = E.v[2] + (u2zf2 - u2.y) * hz;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (add_R) {
        e[c].x += R.v[c];
        e[c].y += R.v[c];
      }
      if (!second) e[c].y = 0.0;
      acc[c] += e[c].x * e[c].x + e[c].y * e[c].y;
      st2(eps.p[c], p.off, e[c]);
    }
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}

// ----------------------------------------------------------------------------- strain + stress from u
// Displacement-based pass for any mixing rule (the laminate rule in particular): eps_k = E + sym grad u_k is
// formed in registers exactly as k_eps_norm does (epsOperatorStaggered  F:18614-18692), its sums of squares
// are accumulated (component_norm  F:10127), and tau = P(eps_k) - C0 : eps_k (calcStress  F:18134-18184) is
// what goes to memory -- the strain is never stored (saves its 6 writes + 6 reads per voxel); the
// divergence follows as its own sweep (the laminate Newton solve is too costly to repeat at six neighbours).
template <int MIX, int NPH>
__global__ __launch_bounds__(kBlock, 2) void k_u_stress(Grid g, StressParams sp, FieldPtrs<3> u, FieldPtrs<kMaxPhases> phi,
                                                        FieldPtrs<3> normals, FieldPtrs<6> tau, Vec6 E, double* partial,
                                                        int* error_flag, Sweep ry) {
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
    const long rowoff = p.off - p.k;
    const int kb = p.k == 0 ? g.nz - 1 : p.k - 1;
    const int kf2 = (p.k + 2 >= g.nz) ? p.k + 2 - g.nz : p.k + 2;

    const double2 u0 = ld2(u.p[0], p.off), u1 = ld2(u.p[1], p.off), u2 = ld2(u.p[2], p.off);
    const double2 u0xf = ld2(u.p[0], p.off + xf), u1xb = ld2(u.p[1], p.off + xb), u2xb = ld2(u.p[2], p.off + xb);
    const double2 u0yb = ld2(u.p[0], p.off + yb), u1yf = ld2(u.p[1], p.off + yf), u2yb = ld2(u.p[2], p.off + yb);
    const double u0zb = u.p[0][rowoff + kb], u1zb = u.p[1][rowoff + kb];
    const double u2zf2 = u.p[2][rowoff + kf2];
    const double u2zf1 = second ? u2.y : u.p[2][rowoff];
    double2 f[NPH], nn[3];
#pragma unroll
    for (int q = 0; q < NPH; ++q) f[q] = q < sp.pt.n ? ld2(phi.p[q], p.off) : make_double2(0.0, 0.0);
    nn[0] = nn[1] = nn[2] = make_double2(0.0, 0.0);
    if (MIX == kMixLaminate) {
      bool mixed = false;
#pragma unroll
      for (int q = 0; q < NPH; ++q)
        mixed = mixed || (f[q].x != 0.0 && f[q].x != 1.0) || (f[q].y != 0.0 && f[q].y != 1.0);
      if (mixed) {
#pragma unroll
        for (int c = 0; c < 3; ++c) nn[c] = ld2(normals.p[c], p.off);
      }
    }

    double2 e[6];
    e[3].x = E.v[3] + 0.5 * ((u2.x - u2yb.x) * hy + (u1.x - u1zb) * hz);
    e[4].x = E.v[4] + 0.5 * ((u2.x - u2xb.x) * hx + (u0.x - u0zb) * hz);
    e[5].x = E.v[5] + 0.5 * ((u1.x - u1xb.x) * hx + (u0.x - u0yb.x) * hy);
    e[0].x = E.v[0] + (u0xf.x - u0.x) * hx;
    e[1].x = E.v[1] + (u1yf.x - u1.x) * hy;
    e[2].x = E.v[2] + (u2zf1 - u2.x) * hz;
    e[3].y = E.v[3] + 0.5 * ((u2.y - u2yb.y) * hy + (u1.y - u1.x) * hz);
    e[4].y = E.v[4] + 0.5 * ((u2.y - u2xb.y) * hx + (u0.y - u0.x) * hz);
    e[5].y = E.v[5] + 0.5 * ((u1.y - u1xb.y) * hx + (u0.y - u0yb.y) * hy);
    e[0].y = E.v[0] + (u0xf.y - u0.y) * hx;
    e[1].y = E.v[1] + (u1yf.y - u1.y) * hy;
    e[2].y = E.v[2] + (u2zf2 - u2.y) * hz;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (!second) e[c].y = 0.0;
      acc[c] += e[c].x * e[c].x + e[c].y * e[c].y;
    }
    double F[6], ph[NPH], nv[3], P0[6], P1[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) F[c] = e[c].x;
#pragma unroll
    for (int q = 0; q < NPH; ++q) ph[q] = f[q].x;
    nv[0] = nn[0].x; nv[1] = nn[1].x; nv[2] = nn[2].x;
    int err = stress_voxel<NPH>(F, ph, nv, sp, P0);
    if (second) {
#pragma unroll
      for (int c = 0; c < 6; ++c) F[c] = e[c].y;
#pragma unroll
      for (int q = 0; q < NPH; ++q) ph[q] = f[q].y;
      nv[0] = nn[0].y; nv[1] = nn[1].y; nv[2] = nn[2].y;
      err |= stress_voxel<NPH>(F, ph, nv, sp, P1);
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) P1[c] = 0.0;
    }
    if (err) atomicOr(error_flag, 1);
#pragma unroll
    for (int c = 0; c < 6; ++c) st2(tau.p[c], p.off, make_double2(P0[c], P1[c]));
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}

// ----------------------------------------------------------------------------- CG in displacement space
// runCGElasticity  F:23153-23247 keeps the strain-like vectors eps, r, p, w.  With prescribed mean strains every one of
// them is a staggered symmetric gradient (eps = E + grad_s u_e; r, p, w = grad_s u_r, u_p, u_w: the operator
// -Gamma0 (C - C0) maps onto such fields), so the solver can carry the 3-component displacements instead of the
// 6-component strains: the vector updates become point-wise on half the data, the inner products (innerProductL2
// F:20955-21038, shear products doubled) evaluate the gradients on the fly.
struct StrainPair {
  double2 e[6];
};

// staggered symmetric gradient of u at the pair p (epsOperatorStaggered  F:18632-18686 without E)
__device__ __forceinline__ StrainPair grad_s_pair(const Grid& g, const FieldPtrs<3>& u, const PairPos& p, bool second) {
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  // x neighbours through Grid::xw_lo / xw_hi: periodic in a whole grid, the spare planes of the neighbours in an x-slab
  const long xf = (p.i + 1 == g.nx ? (long)(g.nx - g.xw_hi) - p.i : 1L) * g.nyzp;
  const long xb = (p.i == 0 ? (long)(g.xw_lo - 1) : -1L) * g.nyzp;
  const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
  const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
  const long rowoff = p.off - p.k;
  const int kb = p.k == 0 ? g.nz - 1 : p.k - 1;
  const int kf2 = (p.k + 2 >= g.nz) ? p.k + 2 - g.nz : p.k + 2;
  const double2 u0 = ld2(u.p[0], p.off), u1 = ld2(u.p[1], p.off), u2 = ld2(u.p[2], p.off);
  const double2 u0xf = ld2(u.p[0], p.off + xf), u1xb = ld2(u.p[1], p.off + xb), u2xb = ld2(u.p[2], p.off + xb);
  const double2 u0yb = ld2(u.p[0], p.off + yb), u1yf = ld2(u.p[1], p.off + yf), u2yb = ld2(u.p[2], p.off + yb);
  const double u0zb = u.p[0][rowoff + kb], u1zb = u.p[1][rowoff + kb];
  const double u2zf2 = u.p[2][rowoff + kf2];
  const double u2zf1 = second ? u2.y : u.p[2][rowoff];
  StrainPair s;
  s.e[3].x = 0.5 * ((u2.x - u2yb.x) * hy + (u1.x - u1zb) * hz);
  s.e[4].x = 0.5 * ((u2.x - u2xb.x) * hx + (u0.x - u0zb) * hz);
  s.e[5].x = 0.5 * ((u1.x - u1xb.x) * hx + (u0.x - u0yb.x) * hy);
  s.e[0].x = (u0xf.x - u0.x) * hx;
  s.e[1].x = (u1yf.x - u1.x) * hy;
  s.e[2].x = (u2zf1 - u2.x) * hz;
  s.e[3].y = 0.5 * ((u2.y - u2yb.y) * hy + (u1.y - u1.x) * hz);
  s.e[4].y = 0.5 * ((u2.y - u2xb.y) * hx + (u0.y - u0.x) * hz);
  s.e[5].y = 0.5 * ((u1.y - u1xb.y) * hx + (u0.y - u0yb.y) * hy);
  s.e[0].y = (u0xf.y - u0.y) * hx;
  s.e[1].y = (u1yf.y - u1.y) * hy;
  s.e[2].y = (u2zf2 - u2.y) * hz;
  if (!second) {
#pragma unroll
    for (int c = 0; c < 6; ++c) s.e[c].y = 0.0;
  }
  return s;
}

// MODE 0:  out[0] = sum grad_s a : (grad_s a - grad_s b)                                  (p : (p - w))
// MODE 1:  out[0..5] = sum (E + grad_s a)_c^2 ,  out[6] = sum grad_s b : grad_s b         (norms of eps, r : r)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_cgu_dot(Grid g, FieldPtrs<3> a, FieldPtrs<3> b, Vec6 E, double* partial, Sweep ry) {
  __shared__ double smem[4 * 7];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx = (run.first + it * run.stride) * kBlock + threadIdx.x;
    if (pidx >= npairs) continue;
    const PairPos p = pair_pos_tiled(pidx, g, ry);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    const StrainPair ea = grad_s_pair(g, a, p, second), eb = grad_s_pair(g, b, p, second);
    if (MODE == 0) {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double wgt = c < 3 ? 1.0 : 2.0;
        sx += wgt * (ea.e[c].x * (ea.e[c].x - eb.e[c].x));
        sy += wgt * (ea.e[c].y * (ea.e[c].y - eb.e[c].y));
      }
      acc[0] += sx + sy;
    } else {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double ex = E.v[c] + ea.e[c].x, ey = second ? E.v[c] + ea.e[c].y : 0.0;
        acc[c] += ex * ex + ey * ey;
        const double wgt = c < 3 ? 1.0 : 2.0;
        sx += wgt * (eb.e[c].x * eb.e[c].x);
        sy += wgt * (eb.e[c].y * eb.e[c].y);
      }
      acc[6] += sx + sy;
    }
  }
  block_reduce<7>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 7; ++c) partial[(long)blockIdx.x * 7 + c] = acc[c];
  }
}

// point-wise vector updates on 3-component fields (padding included: harmless)
//   MODE 0:  x += a y ;  r -= a (y - w)          (u_e += alpha u_p ; u_r -= alpha (u_p - u_w))
//   MODE 1:  y = r + a y                          (u_p = u_r + beta u_p)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_cgu_axpy(long n2, FieldPtrs<3> x, FieldPtrs<3> y, FieldPtrs<3> r, FieldPtrs<3> w,
                                                     const double* sc, int i_num, int i_den, double nvox, double small) {
  // the CG coefficient stays on the device: a = (sc[num] / N + tiny) / (sc[den] / N + tiny) from the sums the dot
  // sweeps left there (alpha = gamma / (p:(p - w)),  beta = delta / gamma  F:23201-23240), no host round trip
  const double a = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (MODE == 0) {
        double2 xv = ld2(x.p[c], 2 * i), rv = ld2(r.p[c], 2 * i);
        const double2 yv = ld2(y.p[c], 2 * i), wv = ld2(w.p[c], 2 * i);
        xv.x = xv.x + a * yv.x;
        xv.y = xv.y + a * yv.y;
        rv.x = rv.x - a * (yv.x - wv.x);
        rv.y = rv.y - a * (yv.y - wv.y);
        st2(x.p[c], 2 * i, xv);
        st2(r.p[c], 2 * i, rv);
      } else {
        double2 yv = ld2(y.p[c], 2 * i);
        const double2 rv = ld2(r.p[c], 2 * i);
        yv.x = rv.x + a * yv.x;
        yv.y = rv.y + a * yv.y;
        st2(y.p[c], 2 * i, yv);
      }
    }
  }
}

// ----------------------------------------------------------------------------- voxel lists of the laminate correction
// Ordered stream compaction for the voxel lists below: a workgroup owns kCompactChunk consecutive voxels, counts its
// flagged voxels (pass 1, offsets == nullptr), and after an exclusive scan of the workgroup counts writes them in
// voxel order (pass 2).  Sorted lists keep the gathers of the per-pass kernels on neighbouring cache lines, and the
// order does not depend on the schedule.
constexpr int kCompactChunk = 16 * kBlock;

// Order of the lists: bricks of 8 x 8 x 8 voxels (z fastest inside a brick, bricks in z, y, x order) instead of plain voxel
// order.  An interface is a surface: in voxel order its voxels of one x plane are thousands of list entries away from their
// x neighbours, in brick order a workgroup's 256 entries are a compact patch of the surface whose members share the rows
// they gather from (the u gather of k_interface_strain fetched 9 separate 64-byte pieces per voxel: 580 B for 150 B of data)
// and whose d values lie next to each other for k_delta_div (512^3: stage 2.40 -> 2.12 ms against voxel order).
struct BrickWalk {
  int nbx, nby, nbz;
  long count;   // padded traversal length
};
inline BrickWalk brick_walk(const Grid& g) {
  BrickWalk w;
  w.nbx = (g.nx + 7) / 8;
  w.nby = (g.ny + 7) / 8;
  w.nbz = (g.nz + 7) / 8;
  w.count = (long)w.nbx * w.nby * w.nbz * 512;
  return w;
}
// traversal position t -> voxel (i, j, k); false: a padding position of a partial brick (or past the end)
__device__ __forceinline__ bool walk_voxel(const BrickWalk& w, const Grid& g, long t, int* i, int* j, int* k) {
  if (t >= w.count) return false;
  const long b = t >> 9;
  const int r = (int)(t & 511);
  const int bk = (int)(b % w.nbz);
  const long b2 = b / w.nbz;
  const int bj = (int)(b2 % w.nby), bi = (int)(b2 / w.nby);
  *i = bi * 8 + (r >> 6);
  *j = bj * 8 + ((r >> 3) & 7);
  *k = bk * 8 + (r & 7);
  return *i < g.nx && *j < g.ny && *k < g.nz;
}

// Per-pass kernels over a list: chunk of 256 entries per workgroup, the chunks of one XCD contiguous in the list (blockIdx
// round-robins over the 8 XCDs) so that neighbouring patches of the surface meet in one L2.
__device__ __forceinline__ unsigned list_chunk() {
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned per = (nb + 7) / 8;
  const unsigned c = (b % 8) * per + b / 8;
  return c;   // may be >= nb for the last XCD's tail: the caller's bounds check on the entry index covers it
}

__device__ __forceinline__ unsigned ordered_slot(bool flag, unsigned& base, unsigned* wcount) {
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wcount[w] = (unsigned)__popcll(m);
  __syncthreads();
  unsigned before = 0, total = 0;
  for (int i = 0; i < kBlock / 64; ++i) {
    const unsigned c = wcount[i];
    if (i < w) before += c;
    total += c;
  }
  const unsigned slot = base + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
  base += total;
  __syncthreads();
  return slot;
}

__global__ __launch_bounds__(kBlock) void k_scan_counts(unsigned* counts, int n) {   // exclusive scan in place, total -> counts[n]
  __shared__ unsigned chunk_sum[kBlock];
  const int per = (n + kBlock - 1) / kBlock;
  const int lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  unsigned run = 0;
  for (int i = lo; i < hi; ++i) run += counts[i];
  chunk_sum[threadIdx.x] = run;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned acc = 0;
    for (int t = 0; t < kBlock; ++t) {
      const unsigned c = chunk_sum[t];
      chunk_sum[t] = acc;
      acc += c;
    }
    counts[n] = acc;
  }
  __syncthreads();
  run = chunk_sum[threadIdx.x];
  for (int i = lo; i < hi; ++i) {
    const unsigned c = counts[i];
    counts[i] = run;
    run += c;
  }
}

__global__ __launch_bounds__(kBlock) void k_mixed_list(Grid g, BrickWalk bw, int nph, FieldPtrs<kMaxPhases> phi, unsigned* list,
                                                       unsigned* counts) {
  __shared__ unsigned wcount[kBlock / 64];
  unsigned base = list ? counts[blockIdx.x] : 0u;
  for (int r = 0; r < kCompactChunk / kBlock; ++r) {
    const long v = (long)blockIdx.x * kCompactChunk + r * kBlock + threadIdx.x;
    bool mixed = false;
    long off = 0;
    int vi, vj, vk;
    if (walk_voxel(bw, g, v, &vi, &vj, &vk)) {
      off = ((long)vi * g.ny + vj) * g.nzp + vk;
      for (int q = 0; q < nph; ++q) {
        const double f = phi.p[q][off];
        mixed = mixed || (f != 0.0 && f != 1.0);
      }
    }
    const unsigned slot = ordered_slot(mixed, base, wcount);
    if (mixed && list) list[slot] = (unsigned)off;
  }
  if (!list && threadIdx.x == 0) counts[blockIdx.x] = base;
}

// Laminate mixing in the displacement loop (u_loop = 2).  The divergence is linear in the polarisation, so
//   f = div tau_voigt + div (tau_laminate - tau_voigt),
// and the second term lives on the interface voxels only: the tiled Voigt sweep runs unchanged over all voxels, then
//   k_interface_*      d_j = P_laminate(eps_j) - P_voigt(eps_j) for every interface voxel j (eps from u by the strain
//                      stencil, the one-step Newton solve of laminate_split), stored compactly [j][6] (see below);
//   k_delta_div        every voxel whose divergence stencil touches an interface voxel adds div d to its f, gathering d
//                      through seven precomputed slots (self, x-1, x+1, y-1, y+1, z-1, z+1; -1 = not an interface voxel).
// One writer per voxel, fixed operation order: deterministic, no atomics.  The lists are built once per geometry.
__global__ __launch_bounds__(kBlock) void k_fill_int(int* p, long n, int v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

__global__ __launch_bounds__(kBlock) void k_mixed_map(const unsigned* list, unsigned n, int* map) {
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) map[list[j]] = (int)j;
}

struct VoxelNeighbours {
  long xf, xb, yf, yb, zf, zb;
};
// x neighbours through Grid::xw_lo / xw_hi: periodic in a whole grid, the spare planes of the neighbours in an x-slab
__device__ __forceinline__ VoxelNeighbours voxel_neighbours(const Grid& g, int i, int j, int k) {
  VoxelNeighbours n;
  n.xf = (i + 1 == g.nx ? (long)(g.nx - g.xw_hi) - i : 1L) * g.nyzp;
  n.xb = (i == 0 ? (long)(g.xw_lo - 1) : -1L) * g.nyzp;
  n.yf = (j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
  n.yb = (j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
  n.zf = (k + 1 == g.nz ? -(long)(g.nz - 1) : 1L);
  n.zb = (k == 0 ? (long)(g.nz - 1) : -1L);
  return n;
}

// aff == nullptr: count only.  slots: 8 ints per entry (self, xb, xf, yb, yf, zb, zf, unused).
__global__ __launch_bounds__(kBlock) void k_affected_list(Grid g, BrickWalk bw, const int* map, unsigned* aff, int* slots,
                                                          unsigned* counts) {
  __shared__ unsigned wcount[kBlock / 64];
  unsigned base = aff ? counts[blockIdx.x] : 0u;
  for (int r = 0; r < kCompactChunk / kBlock; ++r) {
    const long v = (long)blockIdx.x * kCompactChunk + r * kBlock + threadIdx.x;
    bool any = false;
    long off = 0;
    int sl[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int i, j, k;
    if (walk_voxel(bw, g, v, &i, &j, &k)) {
      off = ((long)i * g.ny + j) * g.nzp + k;
      const VoxelNeighbours nb = voxel_neighbours(g, i, j, k);
      // x-slab (xw_hi != nx): the x neighbours of the two boundary planes live on other ranks; their part of the
      // divergence arrives as dense planes (k_delta_div_halo)
      const bool slab = g.xw_hi != g.nx;
      sl[0] = map[off];
      sl[1] = (slab && i == 0) ? -1 : map[off + nb.xb];
      sl[2] = (slab && i + 1 == g.nx) ? -1 : map[off + nb.xf];
      sl[3] = map[off + nb.yb];
      sl[4] = map[off + nb.yf]; sl[5] = map[off + nb.zb]; sl[6] = map[off + nb.zf];
      for (int t = 0; t < 7; ++t) any = any || sl[t] >= 0;
    }
    const unsigned e = ordered_slot(any, base, wcount);
    if (any && aff) {
      aff[e] = (unsigned)off;
      for (int t = 0; t < 8; ++t) slots[(long)e * 8 + t] = sl[t];
    }
  }
  if (!aff && threadIdx.x == 0) counts[blockIdx.x] = base;
}

// ---- laminate correction, compact form ---------------------------------------------------------------------------------
// d_j = tau_laminate - tau_voigt of the interface voxels in two kernels (round 1: one, k_laminate_delta):
//   k_interface_strain  eps_j from u by the strain stencil -- a light gather kernel at full occupancy (the scattered loads
//                       are what costs), compact SoA [6][n] in list order;
//   k_interface_solve   d_j from eps_j, the phase fractions and the normal, all compact and coalesced (static copies made
//                       once per geometry by k_interface_static): the one-step Newton solve of laminate_split, 190 VGPRs,
//                       no scattered loads.
// 512^3, 3.3 M interface voxels: 0.60 ms -> 0.26 + 0.11 ms.  Measured and rejected: d added to the polarisation
// inside the tiled sweep before its neighbour exchange (the dependent gather of d stalls the barrier-locked march: sweep
// 1.85 -> 2.67 ms against 0.53 ms for k_delta_div), and the sweep emitting eps_j itself (mask load + scattered stores:
// sweep 1.85 -> 2.41 ms against 0.26 ms for k_interface_strain).
__global__ __launch_bounds__(kBlock) void k_interface_static(Grid g, int nph, FieldPtrs<kMaxPhases> phi, FieldPtrs<3> normals,
                                                             const unsigned* list, unsigned n, double* phic, double* nrmc) {
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const long off = list[idx];
    for (int q = 0; q < nph; ++q) phic[(long)q * n + idx] = phi.p[q][off];
    for (int c = 0; c < 3; ++c) nrmc[(long)c * n + idx] = normals.p[c][off];
  }
}

__global__ __launch_bounds__(kBlock) void k_interface_strain(Grid g, FieldPtrs<3> u, Vec6 E, const unsigned* list, unsigned n,
                                                             double* epsc) {
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  {
    const unsigned idx = list_chunk() * kBlock + threadIdx.x;
    if (idx >= n) return;
    const long off = list[idx];
    const long row = off / g.nzp;
    const int k = (int)(off - row * g.nzp);
    const int i = (int)(row / g.ny), j = (int)(row - (long)i * g.ny);
    const VoxelNeighbours nb = voxel_neighbours(g, i, j, k);
    const double u0 = u.p[0][off], u1 = u.p[1][off], u2 = u.p[2][off];
    // epsOperatorStaggered  F:18632-18686 for one voxel
    epsc[3L * n + idx] = E.v[3] + 0.5 * ((u2 - u.p[2][off + nb.yb]) * hy + (u1 - u.p[1][off + nb.zb]) * hz);
    epsc[4L * n + idx] = E.v[4] + 0.5 * ((u2 - u.p[2][off + nb.xb]) * hx + (u0 - u.p[0][off + nb.zb]) * hz);
    epsc[5L * n + idx] = E.v[5] + 0.5 * ((u1 - u.p[1][off + nb.xb]) * hx + (u0 - u.p[0][off + nb.yb]) * hy);
    epsc[0L * n + idx] = E.v[0] + (u.p[0][off + nb.xf] - u0) * hx;
    epsc[1L * n + idx] = E.v[1] + (u.p[1][off + nb.yf] - u1) * hy;
    epsc[2L * n + idx] = E.v[2] + (u.p[2][off + nb.zf] - u2) * hz;
  }
}

template <int NPH>
__global__ __launch_bounds__(kBlock) void k_interface_solve(StressParams sp, const double* epsc, const double* phic,
                                                            const double* nrmc, unsigned n, double* dtau, int* error_flag) {
  {
    const unsigned idx = list_chunk() * kBlock + threadIdx.x;
    if (idx >= n) return;
    double F[6], ph[NPH], nv[3], Pl[6], Pv[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) F[c] = epsc[(long)c * n + idx];
#pragma unroll
    for (int q = 0; q < NPH; ++q) ph[q] = q < sp.pt.n ? phic[(long)q * n + idx] : 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) nv[c] = nrmc[(long)c * n + idx];
    // the reference-medium part of the polarisation is the same in both and cancels
    if (pk1_laminate<NPH>(F, ph, nv, sp.pt, sp.alpha, false, sp.eps_g, sp.eps_a, Pl)) atomicOr(error_flag, 1);
    pk1_voigt<NPH>(F, ph, sp.pt, sp.alpha, false, Pv);
#pragma unroll
    for (int c = 0; c < 6; ++c) dtau[(long)idx * 6 + c] = Pl[c] - Pv[c];
  }
}

__global__ __launch_bounds__(kBlock) void k_delta_div(Grid g, const unsigned* aff, const int* slots, unsigned n,
                                                      const double* dtau, FieldPtrs<3> f) {
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  {
    const unsigned e = list_chunk() * kBlock + threadIdx.x;
    if (e >= n) return;
    const long off = aff[e];
    const int4 sa = reinterpret_cast<const int4*>(slots)[2 * (long)e];       // self, xb, xf, yb
    const int4 sb = reinterpret_cast<const int4*>(slots)[2 * (long)e + 1];   // yf, zb, zf, -
    auto d = [&](int slot, int c) { return slot < 0 ? 0.0 : dtau[(long)slot * 6 + c]; };
    // divOperatorStaggered  F:18853-18908 applied to the difference field
    const double d0 = (d(sa.x, 0) - d(sa.y, 0)) * hx + (d(sb.x, 5) - d(sa.x, 5)) * hy + (d(sb.z, 4) - d(sa.x, 4)) * hz;
    const double d1 = (d(sa.z, 5) - d(sa.x, 5)) * hx + (d(sa.x, 1) - d(sa.w, 1)) * hy + (d(sb.z, 3) - d(sa.x, 3)) * hz;
    const double d2 = (d(sa.z, 4) - d(sa.x, 4)) * hx + (d(sb.x, 3) - d(sa.x, 3)) * hy + (d(sa.x, 2) - d(sb.y, 2)) * hz;
    f.p[0][off] += d0;
    f.p[1][off] += d1;
    f.p[2][off] += d2;
  }
}

// x-slabs: the difference field of the two boundary planes as dense planes for the neighbours (the shape of the
// polarisation halo of the strain-state pipeline): lo = (d5, d4) of the first plane -> left neighbour (its x+1 terms),
// hi = d0 of the last plane -> right neighbour (its x-1 term).  The planes are zeroed before; the interface voxels of a
// boundary plane are few, every list entry looks at its plane only.
__global__ __launch_bounds__(kBlock) void k_delta_pack(Grid g, const unsigned* list, unsigned n, const double* dtau, double* lo,
                                                       double* hi) {
  const long last = (long)(g.nx - 1) * g.nyzp;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const long off = list[idx];
    if (off < g.nyzp) {
      lo[off] = dtau[(long)idx * 6 + 5];
      lo[g.nyzp + off] = dtau[(long)idx * 6 + 4];
    }
    if (off >= last) hi[off - last] = dtau[(long)idx * 6 + 0];
  }
}

// ... and their terms of divOperatorStaggered F:18853-18908 on the receiving side: f0(0,j,k) -= d0(-1,j,k) hx,
// f1(nx-1,j,k) += d5(nx,j,k) hx, f2(nx-1,j,k) += d4(nx,j,k) hx
__global__ __launch_bounds__(kBlock) void k_delta_div_halo(Grid g, const double* from_lo, const double* from_hi, FieldPtrs<3> f) {
  const long last = (long)(g.nx - 1) * g.nyzp;
  const double hx = g.hx;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < g.nyzp; idx += (long)gridDim.x * blockDim.x) {
    const double a = from_lo[idx], b = from_hi[idx], c = from_hi[g.nyzp + idx];
    if (a != 0.0) f.p[0][idx] -= a * hx;
    if (b != 0.0) f.p[1][last + idx] += b * hx;
    if (c != 0.0) f.p[2][last + idx] += c * hx;
  }
}

// out[c] += in[c], c < n (a handful of device scalars)
__global__ void k_add_small(double* out, const double* in, int n) {
  if ((int)threadIdx.x < n) out[threadIdx.x] += in[threadIdx.x];
}

// ----------------------------------------------------------------------------- viscosity: strain + Delta-operator tail
// DeltaOperatorStaggered  F:20438-20452 after the Green operator:  eta = (E - coef <tau>) + sym grad u + coef tau  with
// coef = 2 alpha / (4 mu0), the mean <tau> = tau_sum / N read from device memory (no host round trip), and the sums of
// squares of eta.  Same operation order as k_eps_norm followed by xpay.
// NPH > 0: no stored tau; `tau` holds the strain the pass started from and the polarisation is re-evaluated from it
// (a point-wise function of that strain: same arithmetic as k_stress, so the same values).
template <int NPH>
__global__ __launch_bounds__(kBlock) void k_eps_delta(Grid g, FieldPtrs<3> u, FieldPtrs<6> tau, const double* tau_sum,
                                                      double nvox, Vec6 E, double coef, FieldPtrs<6> eps, double* partial,
                                                      StressParams sp, FieldPtrs<kMaxPhases> phi, Sweep ry) {
  __shared__ double smem[4 * 6];
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  double adj[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) adj[c] = E.v[c] - coef * (tau_sum[c] / nvox);
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx = (run.first + it * run.stride) * kBlock + threadIdx.x;
    if (pidx >= npairs) continue;
    const PairPos p = pair_pos_tiled(pidx, g, ry);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    // x neighbours through Grid::xw_lo / xw_hi: periodic in a whole grid, the spare planes of the neighbours in an x-slab
    const long xf = (p.i + 1 == g.nx ? (long)(g.nx - g.xw_hi) - p.i : 1L) * g.nyzp;
    const long xb = (p.i == 0 ? (long)(g.xw_lo - 1) : -1L) * g.nyzp;
    const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
    const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
    const long rowoff = p.off - p.k;
    const int kb = p.k == 0 ? g.nz - 1 : p.k - 1;
    const int kf2 = (p.k + 2 >= g.nz) ? p.k + 2 - g.nz : p.k + 2;
    const double2 u0 = ld2(u.p[0], p.off), u1 = ld2(u.p[1], p.off), u2 = ld2(u.p[2], p.off);
    const double2 u0xf = ld2(u.p[0], p.off + xf), u1xb = ld2(u.p[1], p.off + xb), u2xb = ld2(u.p[2], p.off + xb);
    const double2 u0yb = ld2(u.p[0], p.off + yb), u1yf = ld2(u.p[1], p.off + yf), u2yb = ld2(u.p[2], p.off + yb);
    const double u0zb = u.p[0][rowoff + kb], u1zb = u.p[1][rowoff + kb];
    const double u2zf2 = u.p[2][rowoff + kf2];
    const double u2zf1 = second ? u2.y : u.p[2][rowoff];
    double2 e[6];
    e[3].x = adj[3] + 0.5 * ((u2.x - u2yb.x) * hy + (u1.x - u1zb) * hz);
    e[4].x = adj[4] + 0.5 * ((u2.x - u2xb.x) * hx + (u0.x - u0zb) * hz);
    e[5].x = adj[5] + 0.5 * ((u1.x - u1xb.x) * hx + (u0.x - u0yb.x) * hy);
    e[0].x = adj[0] + (u0xf.x - u0.x) * hx;
    e[1].x = adj[1] + (u1yf.x - u1.x) * hy;
    e[2].x = adj[2] + (u2zf1 - u2.x) * hz;
    e[3].y = adj[3] + 0.5 * ((u2.y - u2yb.y) * hy + (u1.y - u1.x) * hz);
    e[4].y = adj[4] + 0.5 * ((u2.y - u2xb.y) * hx + (u0.y - u0.x) * hz);
    e[5].y = adj[5] + 0.5 * ((u1.y - u1xb.y) * hx + (u0.y - u0yb.y) * hy);
    e[0].y = adj[0] + (u0xf.y - u0.y) * hx;
    e[1].y = adj[1] + (u1yf.y - u1.y) * hy;
    e[2].y = adj[2] + (u2zf2 - u2.y) * hz;
    double2 tt[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) tt[c] = ld2(tau.p[c], p.off);
    if constexpr (NPH > 0) {
      double2 fc[NPH > 0 ? NPH : 1];
      double ph[NPH > 0 ? NPH : 1];
#pragma unroll
      for (int q = 0; q < NPH; ++q) fc[q] = q < sp.pt.n ? ld2(phi.p[q], p.off) : make_double2(0.0, 0.0);
      double2 o[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) o[c] = tt[c];
#pragma unroll
      for (int q = 0; q < NPH; ++q) ph[q] = fc[q].x;
#pragma unroll
      for (int c = 0; c < 3; ++c) tt[c].x = voigt_tau_normal<NPH>(o[c].x, o[0].x, o[1].x, o[2].x, ph, sp);
#pragma unroll
      for (int c = 3; c < 6; ++c) tt[c].x = voigt_tau_shear<NPH>(o[c].x, ph, sp);
#pragma unroll
      for (int q = 0; q < NPH; ++q) ph[q] = fc[q].y;
#pragma unroll
      for (int c = 0; c < 3; ++c) tt[c].y = voigt_tau_normal<NPH>(o[c].y, o[0].y, o[1].y, o[2].y, ph, sp);
#pragma unroll
      for (int c = 3; c < 6; ++c) tt[c].y = voigt_tau_shear<NPH>(o[c].y, ph, sp);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      e[c].x = e[c].x + coef * tt[c].x;
      e[c].y = second ? e[c].y + coef * tt[c].y : 0.0;
      acc[c] += e[c].x * e[c].x + e[c].y * e[c].y;
      st2(eps.p[c], p.off, e[c]);
    }
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}

// ----------------------------------------------------------------------------- plain reductions
// Per-component sums (TensorField::average F:10171-10210) or sums of squares over the
// valid voxels of NC components.
template <int NC, bool SQUARE>
__global__ __launch_bounds__(kBlock) void k_sum(Grid g, FieldPtrs<NC> x, double* partial) {
  __shared__ double smem[4 * NC];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double2 v = ld2(x.p[c], p.off);
      if (!second) v.y = 0.0;
      acc[c] += SQUARE ? (v.x * v.x + v.y * v.y) : (v.x + v.y);
    }
  }
  block_reduce<NC>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) partial[(long)blockIdx.x * NC + c] = acc[c];
  }
}

// ----------------------------------------------------------------------------- CG vector kernels
// runCGElasticity  F:23153-23247.  One kernel per vector update, each fused with the reduction that
// follows it in the algorithm.  MODE:
//   0  r <- r + (E - eps)                      [adjustResidual F:10012]       sum r:r
//   1  (no update)                                                             sum p:(p - w)
//   2  eps <- eps + a*p                        [xpay F:9819]                   per-component sum eps_c^2
//   3  r <- r + a*(p - w)   (a = -alpha)       [xpaymz F:9993]                 sum r:r
//   4  p <- r + a*p         (a = beta)         [xpay]
// ":" is innerProductL2  F:20955-21038: shear products doubled.  x, y, z are 6-component fields.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_cg(Grid g, FieldPtrs<6> x, FieldPtrs<6> y, FieldPtrs<6> z, Vec6 E, double a,
                                               double* partial) {
  __shared__ double smem[4 * 6];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    double2 d[6];  // per component: the two terms of the weighted product (or eps for MODE 2)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (MODE == 0) {  // x = r, y = eps
        double2 r = ld2(x.p[c], p.off);
        const double2 e = ld2(y.p[c], p.off);
        r.x += E.v[c] - e.x;
        r.y += E.v[c] - e.y;
        st2(x.p[c], p.off, r);
        d[c] = make_double2(r.x * r.x, r.y * r.y);
      } else if (MODE == 1) {  // x = p, y = w
        const double2 pp = ld2(x.p[c], p.off), w = ld2(y.p[c], p.off);
        d[c] = make_double2(pp.x * (pp.x - w.x), pp.y * (pp.y - w.y));
      } else if (MODE == 2) {  // x = eps, y = p
        double2 e = ld2(x.p[c], p.off);
        const double2 pp = ld2(y.p[c], p.off);
        e.x = e.x + a * pp.x;
        e.y = e.y + a * pp.y;
        st2(x.p[c], p.off, e);
        d[c] = e;
      } else if (MODE == 3) {  // x = r, y = p, z = w
        double2 r = ld2(x.p[c], p.off);
        const double2 pp = ld2(y.p[c], p.off), w = ld2(z.p[c], p.off);
        r.x = r.x + a * (pp.x - w.x);
        r.y = r.y + a * (pp.y - w.y);
        st2(x.p[c], p.off, r);
        d[c] = make_double2(r.x * r.x, r.y * r.y);
      } else {  // x = p, y = r
        double2 pp = ld2(x.p[c], p.off);
        const double2 r = ld2(y.p[c], p.off);
        pp.x = r.x + a * pp.x;
        pp.y = r.y + a * pp.y;
        st2(x.p[c], p.off, pp);
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[c] += d[c].x * d[c].x + (second ? d[c].y * d[c].y : 0.0);
    } else if (MODE != 4) {
      const double sx = d[0].x + d[1].x + d[2].x + 2 * (d[3].x + d[4].x + d[5].x);
      const double sy = d[0].y + d[1].y + d[2].y + 2 * (d[3].y + d[4].y + d[5].y);
      acc[0] += sx + (second ? sy : 0.0);
    }
  }
  if (MODE != 4) {
    block_reduce<6>(acc, smem, OpSum());
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
    }
  }
}

// eps <- constant tensor E  (setConstant  F:10026)
__global__ __launch_bounds__(kBlock) void k_set_const6(long n2, FieldPtrs<6> x, Vec6 E) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 6; ++c) st2(x.p[c], 2 * i, make_double2(E.v[c], E.v[c]));
  }
}

// min / max of the tangent spectrum over all voxels (reference-material scan)
__global__ __launch_bounds__(kBlock) void k_tangent_minmax(Grid g, PhaseTable pt, int mixing, FieldPtrs<kMaxPhases> phi,
                                                           double* partial, int* error_flag) {
  __shared__ double smem[4 * 2];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[2] = {1.0 / 0.0, 1.0 / 0.0};  // (min, -max) so one OpMin tree serves both
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    for (int s = 0; s < 2; ++s) {
      if (p.k + s >= g.nz) continue;
      double ph[kMaxPhases];
#pragma unroll
      for (int q = 0; q < kMaxPhases; ++q) ph[q] = q < pt.n ? phi.p[q][p.off + s] : 0.0;
      double lo, hi;
      if (tangent_eigs<kMaxPhases>(ph, pt, mixing, &lo, &hi) != 0) {
        atomicOr(error_flag, 1);
        continue;
      }
      acc[0] = acc[0] < lo ? acc[0] : lo;
      acc[1] = acc[1] < -hi ? acc[1] : -hi;
    }
  }
  block_reduce<2>(acc, smem, OpMin());
  if (threadIdx.x == 0) {
    partial[(long)blockIdx.x * 2 + 0] = acc[0];
    partial[(long)blockIdx.x * 2 + 1] = acc[1];
  }
}

// copy one x-plane of a padded component (halo packing)
__global__ __launch_bounds__(kBlock) void k_copy(const double* src, double* dst, long n2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
    st2(dst, 2 * i, ld2(src, 2 * i));
}

// calcStressConst  F:17973-18020 : tau = 2 mu0 eps + lambda0 tr(eps) I  (whole padded arrays)
__global__ __launch_bounds__(kBlock) void k_stress_const(long n2, double two_mu, double lambda, FieldPtrs<6> eps,
                                                         FieldPtrs<6> tau) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    double2 e[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) e[c] = ld2(eps.p[c], 2 * i);
    const double lx = lambda * (e[0].x + e[1].x + e[2].x);
    const double ly = lambda * (e[0].y + e[1].y + e[2].y);
#pragma unroll
    for (int c = 0; c < 3; ++c) st2(tau.p[c], 2 * i, make_double2(e[c].x * two_mu + lx, e[c].y * two_mu + ly));
#pragma unroll
    for (int c = 3; c < 6; ++c) st2(tau.p[c], 2 * i, make_double2(e[c].x * two_mu, e[c].y * two_mu));
  }
}

int grid_for(long nwork, int max_blocks) {
  long b = (nwork + kBlock - 1) / kBlock;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

namespace {
template <int REDUCE>
void stress_dispatch(dim3 grid, const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps,
                     const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& normals, const FieldPtrs<6>& tau,
                     double* partial, int* error_flag, hipStream_t s) {
  const bool lam = sp.mixing == kMixLaminate;
  const bool small = sp.pt.n <= 2;
#define FG_LAUNCH(MIX, NPH)                                                                                        \
  hipLaunchKernelGGL((k_stress<REDUCE, MIX, NPH>), grid, dim3(kBlock), 0, s, g, sp, eps, phi, normals, tau, partial, \
                     error_flag)
  if (lam && small) FG_LAUNCH(kMixLaminate, 2);
  else if (lam) FG_LAUNCH(kMixLaminate, kMaxPhases);
  else if (small) FG_LAUNCH(kMixVoigt, 2);
  else FG_LAUNCH(kMixVoigt, kMaxPhases);
#undef FG_LAUNCH
  FG_HIP_CHECK(hipGetLastError());
}
}  // namespace

long partial_rows(const Grid& g) {
  const long nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  return nb + (nb + 511) / 512 + 8;  // sweep partials + one intermediate fold level
}

int reduce_blocks(const Grid& g) { return grid_for((long)g.nx * g.ny * g.nzc, kMaxReduceBlocks); }

void launch_stress(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps, const FieldPtrs<kMaxPhases>& phi,
                   const FieldPtrs<3>& normals, const FieldPtrs<6>& tau, int* error_flag, hipStream_t s) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  stress_dispatch<0>(dim3(grid_for(npairs, 1 << 20)), g, sp, eps, phi, normals, tau, nullptr, error_flag, s);
}

void launch_stress_mean(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps,
                        const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& normals, double* partial, double* out6,
                        int* error_flag, hipStream_t s) {
  const int nb = reduce_blocks(g);
  FieldPtrs<6> none = {};
  stress_dispatch<1>(dim3(nb), g, sp, eps, phi, normals, none, partial, error_flag, s);
  hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, nb, 6, 0.0, out6);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_energy_mean(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps, const FieldPtrs<kMaxPhases>& phi,
                        const FieldPtrs<3>& normals, double* partial, double* out6, int* error_flag, hipStream_t s) {
  const int nb = reduce_blocks(g);
  FieldPtrs<6> none = {};
  stress_dispatch<2>(dim3(nb), g, sp, eps, phi, normals, none, partial, error_flag, s);
  hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, nb, 6, 0.0, out6);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_stress_const(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<6>& eps, const FieldPtrs<6>& tau,
                         hipStream_t s) {
  const long n2 = g.n / 2;
  hipLaunchKernelGGL(k_stress_const, dim3(grid_for(n2, 1 << 20)), dim3(kBlock), 0, s, n2, 2 * mu_0, lambda_0, eps, tau);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_stress_div_voigt(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps,
                            const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& f, hipStream_t s) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  if (sp.pt.n <= 2)
    hipLaunchKernelGGL((k_stress_div_voigt<2, false>), dim3(sweep_blocks(npairs)), dim3(kBlock), 0, s, g, sp, eps, phi, f,
                       (double*)nullptr, chunk_rows(g));
  else
    hipLaunchKernelGGL((k_stress_div_voigt<kMaxPhases, false>), dim3(sweep_blocks(npairs)), dim3(kBlock), 0, s, g, sp, eps,
                       phi, f, (double*)nullptr, chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
}

void launch_stress_div_sum_voigt(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps,
                                const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& f, double* partial, double* sum6,
                                hipStream_t s) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  const int nb = sweep_blocks(npairs);
  if (sp.pt.n <= 2)
    hipLaunchKernelGGL((k_stress_div_voigt<2, true>), dim3(nb), dim3(kBlock), 0, s, g, sp, eps, phi, f, partial,
                       chunk_rows(g));
  else
    hipLaunchKernelGGL((k_stress_div_voigt<kMaxPhases, true>), dim3(nb), dim3(kBlock), 0, s, g, sp, eps, phi, f, partial,
                       chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sum6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_u_stress_div_voigt(const Grid& g, const StressParams& sp, const FieldPtrs<3>& u,
                               const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& f, const Vec6& E, double* partial,
                               double* sumsq6, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  if (sp.pt.n <= 2)
    hipLaunchKernelGGL((k_u_stress_div_voigt<2>), dim3(nb), dim3(kBlock), 0, s, g, sp, u, phi, f, E, partial,
                       chunk_rows(g));
  else
    hipLaunchKernelGGL((k_u_stress_div_voigt<kMaxPhases>), dim3(nb), dim3(kBlock), 0, s, g, sp, u, phi, f, E, partial,
                       chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_u_stress(const Grid& g, const StressParams& sp, const FieldPtrs<3>& u, const FieldPtrs<kMaxPhases>& phi,
                     const FieldPtrs<3>& normals, const FieldPtrs<6>& tau, const Vec6& E, double* partial, double* sumsq6,
                     int* error_flag, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  const Sweep sw = chunk_rows(g);
  const bool two = sp.pt.n <= 2;
#define FG_LAUNCH(MIX, NPH) \
  hipLaunchKernelGGL((k_u_stress<MIX, NPH>), dim3(nb), dim3(kBlock), 0, s, g, sp, u, phi, normals, tau, E, partial, error_flag, sw)
  if (sp.mixing == kMixLaminate) {
    if (two) FG_LAUNCH(kMixLaminate, 2);
    else FG_LAUNCH(kMixLaminate, kMaxPhases);
  } else {
    if (two) FG_LAUNCH(kMixVoigt, 2);
    else FG_LAUNCH(kMixVoigt, kMaxPhases);
  }
#undef FG_LAUNCH
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_cgu_dot(int mode, const Grid& g, const FieldPtrs<3>& a, const FieldPtrs<3>& b, const Vec6& E, double* partial,
                    double* out7, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  const Sweep sw = chunk_rows(g);
  if (mode == 0) hipLaunchKernelGGL((k_cgu_dot<0>), dim3(nb), dim3(kBlock), 0, s, g, a, b, E, partial, sw);
  else hipLaunchKernelGGL((k_cgu_dot<1>), dim3(nb), dim3(kBlock), 0, s, g, a, b, E, partial, sw);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 7, out7, s);
  FG_HIP_CHECK(hipGetLastError());
}

// The same updates OUT OF PLACE on `count` doubles per component from offset `off` on (x-slabs with the fused CG sweeps: the tile
// kernels write the own planes of the alternate buffers, this one their spare planes -- the halo planes stay valid without an
// exchange because the updates are point-wise with coefficients every rank forms from the same all-reduced sums).
//   MODE 0:  xo = x + a y ;  ro = r - a (y - w)          MODE 1:  xo = r + a y
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_cgu_axpy_oop(long n2, long off, FieldPtrs<3> x, FieldPtrs<3> y, FieldPtrs<3> r,
                                                         FieldPtrs<3> w, FieldPtrs<3> xo, FieldPtrs<3> ro, const double* sc, int i_num,
                                                         int i_den, double nvox, double small) {
  const double a = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    const long o = off + 2 * i;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double2 yv = ld2(y.p[c], o), rv = ld2(r.p[c], o);
      if (MODE == 0) {
        const double2 xv = ld2(x.p[c], o), wv = ld2(w.p[c], o);
        st2(xo.p[c], o, make_double2(xv.x + a * yv.x, xv.y + a * yv.y));
        st2(ro.p[c], o, make_double2(rv.x - a * (yv.x - wv.x), rv.y - a * (yv.y - wv.y)));
      } else {
        st2(xo.p[c], o, make_double2(rv.x + a * yv.x, rv.y + a * yv.y));
      }
    }
  }
}

void launch_cgu_axpy_oop(int mode, const FieldPtrs<3>& x, const FieldPtrs<3>& y, const FieldPtrs<3>& r, const FieldPtrs<3>& w,
                         const FieldPtrs<3>& xo, const FieldPtrs<3>& ro, const double* sc, int i_num, int i_den, double nvox,
                         double small, long off, long count, hipStream_t s) {
  const long n2 = count / 2;
  if (n2 <= 0) return;
  const dim3 grid(grid_for(n2, 1 << 16));
  if (mode == 0) hipLaunchKernelGGL((k_cgu_axpy_oop<0>), grid, dim3(kBlock), 0, s, n2, off, x, y, r, w, xo, ro, sc, i_num, i_den, nvox, small);
  else hipLaunchKernelGGL((k_cgu_axpy_oop<1>), grid, dim3(kBlock), 0, s, n2, off, x, y, r, w, xo, ro, sc, i_num, i_den, nvox, small);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_cgu_axpy(int mode, const Grid& g, const FieldPtrs<3>& x, const FieldPtrs<3>& y, const FieldPtrs<3>& r,
                     const FieldPtrs<3>& w, const double* sc, int i_num, int i_den, double nvox, double small, hipStream_t s,
                     long count) {
  const long n2 = (count > 0 ? count : g.n) / 2;
  const dim3 grid(grid_for(n2, 1 << 16));
  if (mode == 0) hipLaunchKernelGGL((k_cgu_axpy<0>), grid, dim3(kBlock), 0, s, n2, x, y, r, w, sc, i_num, i_den, nvox, small);
  else hipLaunchKernelGGL((k_cgu_axpy<1>), grid, dim3(kBlock), 0, s, n2, x, y, r, w, sc, i_num, i_den, nvox, small);
  FG_HIP_CHECK(hipGetLastError());
}

namespace {
int compact_blocks(const Grid& g) { return (int)((brick_walk(g).count + kCompactChunk - 1) / kCompactChunk); }
// one chunk of kBlock entries per workgroup, rounded up to a multiple of 8 (list_chunk deals whole runs to the XCDs)
dim3 list_grid(unsigned n) {
  const long nb = ((long)n + kBlock - 1) / kBlock;
  return dim3((unsigned)(((nb + 7) / 8) * 8));
}
unsigned scan_counts(unsigned* counts, int nb, hipStream_t s) {
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(kBlock), 0, s, counts, nb);
  FG_HIP_CHECK(hipGetLastError());
  unsigned total = 0;
  FG_HIP_CHECK(hipMemcpyAsync(&total, counts + nb, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  FG_HIP_CHECK(hipStreamSynchronize(s));
  return total;
}
}  // namespace

unsigned launch_mixed_list(const Grid& g, int nph, const FieldPtrs<kMaxPhases>& phi, unsigned** list, hipStream_t s) {
  if ((long)g.nx * g.ny * g.nzp >= (1L << 31)) throw std::runtime_error("grid too large for the 32-bit interface list");
  const int nb = compact_blocks(g);
  unsigned* counts = nullptr;
  FG_HIP_CHECK(hipMalloc(&counts, ((size_t)nb + 1) * sizeof(unsigned)));
  const BrickWalk bw = brick_walk(g);
  hipLaunchKernelGGL(k_mixed_list, dim3(nb), dim3(kBlock), 0, s, g, bw, nph, phi, (unsigned*)nullptr, counts);
  const unsigned n = scan_counts(counts, nb, s);
  *list = nullptr;
  if (n) {
    FG_HIP_CHECK(hipMalloc(list, (size_t)n * sizeof(unsigned)));
    hipLaunchKernelGGL(k_mixed_list, dim3(nb), dim3(kBlock), 0, s, g, bw, nph, phi, *list, counts);
    FG_HIP_CHECK(hipGetLastError());
    FG_HIP_CHECK(hipStreamSynchronize(s));
  }
  FG_HIP_CHECK(hipFree(counts));
  return n;
}

unsigned launch_affected_list(const Grid& g, const unsigned* list, unsigned n, unsigned** aff, int** slots, hipStream_t s) {
  // map: scratch of g.n ints (element offset -> index in the interface list or -1), needed during the build only
  const int nb = compact_blocks(g);
  int* map = nullptr;
  unsigned* counts = nullptr;
  FG_HIP_CHECK(hipMalloc(&map, (size_t)g.n * sizeof(int)));
  FG_HIP_CHECK(hipMalloc(&counts, ((size_t)nb + 1) * sizeof(unsigned)));
  hipLaunchKernelGGL(k_fill_int, dim3(grid_for(g.n, 1 << 16)), dim3(kBlock), 0, s, map, g.n, -1);
  if (n) hipLaunchKernelGGL(k_mixed_map, dim3(grid_for((long)n, 1 << 16)), dim3(kBlock), 0, s, list, n, map);
  const BrickWalk bw = brick_walk(g);
  hipLaunchKernelGGL(k_affected_list, dim3(nb), dim3(kBlock), 0, s, g, bw, map, (unsigned*)nullptr, (int*)nullptr, counts);
  const unsigned m = scan_counts(counts, nb, s);
  *aff = nullptr;
  *slots = nullptr;
  if (m) {
    FG_HIP_CHECK(hipMalloc(aff, (size_t)m * sizeof(unsigned)));
    FG_HIP_CHECK(hipMalloc(slots, (size_t)m * 8 * sizeof(int)));
    hipLaunchKernelGGL(k_affected_list, dim3(nb), dim3(kBlock), 0, s, g, bw, map, *aff, *slots, counts);
    FG_HIP_CHECK(hipGetLastError());
    FG_HIP_CHECK(hipStreamSynchronize(s));
  }
  FG_HIP_CHECK(hipFree(map));
  FG_HIP_CHECK(hipFree(counts));
  return m;
}

void launch_interface_static(const Grid& g, int nph, const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& normals,
                             const unsigned* list, unsigned n, double* phic, double* nrmc, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_interface_static, dim3(grid_for((long)n, 1 << 16)), dim3(kBlock), 0, s, g, nph, phi, normals, list, n, phic,
                     nrmc);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_interface_delta(const Grid& g, const StressParams& sp, const FieldPtrs<3>& u, const Vec6& E, const unsigned* list,
                            unsigned n, double* epsc, const double* phic, const double* nrmc, double* dtau, int* error_flag,
                            hipStream_t s) {
  if (n == 0) return;
  const dim3 grid = list_grid(n);
  hipLaunchKernelGGL(k_interface_strain, grid, dim3(kBlock), 0, s, g, u, E, list, n, epsc);
  if (sp.pt.n <= 2) hipLaunchKernelGGL((k_interface_solve<2>), grid, dim3(kBlock), 0, s, sp, epsc, phic, nrmc, n, dtau, error_flag);
  else hipLaunchKernelGGL((k_interface_solve<kMaxPhases>), grid, dim3(kBlock), 0, s, sp, epsc, phic, nrmc, n, dtau, error_flag);
  FG_HIP_CHECK(hipGetLastError());
}

// six sums over the compact difference array [n][6] (mixed boundary conditions: <tau_laminate> = <tau_voigt> + this / N)
__global__ __launch_bounds__(kBlock) void k_sum_dtau(const double* dtau, unsigned n, double* partial) {
  __shared__ double smem[4 * 6];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] += dtau[(long)j * 6 + c];
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}

void launch_sum_dtau(const double* dtau, unsigned n, double* partial, double* out6, hipStream_t s) {
  const int nb = n ? (int)std::min<long>(((long)n + kBlock - 1) / kBlock, 1024) : 1;
  hipLaunchKernelGGL(k_sum_dtau, dim3(nb), dim3(kBlock), 0, s, dtau, n, partial);
  FG_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, nb, 6, 0.0, out6);
  FG_HIP_CHECK(hipGetLastError());
}

// out = sum_i w[i] in[i], i < n <= 8, over ndoubles values (load-step extrapolation: the polynomial through the fields of
// the last steps evaluated at the new parameter, extrapolateLoadstepPolynomial F:21468-21514 -- the reference forms the
// coefficients per voxel, p = V^-1 f, and then sum_i t^i p_i; with w = V^-T tpowers that is the same linear combination)
struct Lincomb {
  const double* in[8];
  double w[8];
  int n;
};
__global__ __launch_bounds__(kBlock) void k_lincomb(Lincomb a, double* out, long ndoubles) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ndoubles; i += (long)gridDim.x * blockDim.x) {
    double v = 0.0;
    for (int q = 0; q < a.n; ++q) v += a.w[q] * a.in[q][i];
    out[i] = v;
  }
}
void launch_lincomb(int n, const double* const* in, const double* w, double* out, long ndoubles, hipStream_t s) {
  if (n < 1 || n > 8) throw std::runtime_error("launch_lincomb: 1..8 terms");
  Lincomb a;
  a.n = n;
  for (int q = 0; q < 8; ++q) a.in[q] = q < n ? in[q] : nullptr, a.w[q] = q < n ? w[q] : 0.0;
  hipLaunchKernelGGL(k_lincomb, dim3(grid_for(ndoubles, 1 << 16)), dim3(kBlock), 0, s, a, out, ndoubles);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_delta_pack(const Grid& g, const unsigned* list, unsigned n, const double* dtau, double* lo2, double* hi1,
                       hipStream_t s) {
  FG_HIP_CHECK(hipMemsetAsync(lo2, 0, 2 * (size_t)g.nyzp * sizeof(double), s));
  FG_HIP_CHECK(hipMemsetAsync(hi1, 0, (size_t)g.nyzp * sizeof(double), s));
  if (n == 0) return;
  hipLaunchKernelGGL(k_delta_pack, dim3(grid_for((long)n, 1 << 16)), dim3(kBlock), 0, s, g, list, n, dtau, lo2, hi1);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_delta_div_halo(const Grid& g, const double* from_lo1, const double* from_hi2, const FieldPtrs<3>& f, hipStream_t s) {
  hipLaunchKernelGGL(k_delta_div_halo, dim3(grid_for(g.nyzp, 1 << 16)), dim3(kBlock), 0, s, g, from_lo1, from_hi2, f);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_add_small(double* out, const double* in, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_add_small, dim3(1), dim3(64), 0, s, out, in, n);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_delta_div(const Grid& g, const unsigned* aff, const int* slots, unsigned n, const double* dtau,
                      const FieldPtrs<3>& f, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_delta_div, list_grid(n), dim3(kBlock), 0, s, g, aff, slots, n, dtau, f);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_div(const Grid& g, const FieldPtrs<6>& tau, const FieldPtrs<3>& f, const XHalo& h, hipStream_t s) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  hipLaunchKernelGGL(k_div, dim3(sweep_blocks(npairs)), dim3(kBlock), 0, s, g, tau, f, h, chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
}

void launch_g0(const Grid& g, const FieldPtrs<3>& fh, const G0Tables& tb, double c10, double c20, const G0Layout& lay,
               hipStream_t s) {
  const long nfreq = lay.transposed ? (long)lay.nyl * g.nx * g.nzc : (long)g.nx * g.ny * g.nzc;
  hipLaunchKernelGGL(k_g0, dim3(grid_for(nfreq, 1 << 20)), dim3(kBlock), 0, s, g, fh, tb, c10, c20, lay);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_gamma_collocated(const Grid& g, const FieldPtrs<6>& th, const XiTables& xt, double c10, double c20, double beta,
                             const Vec6& E, hipStream_t s) {
  const long nfreq = (long)g.nx * g.ny * g.nzc;
  hipLaunchKernelGGL(k_gamma_collocated, dim3(grid_for(nfreq, 1 << 20)), dim3(kBlock), 0, s, g, th, xt, c10, c20, beta, E);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_eps_norm(const Grid& g, const FieldPtrs<3>& u, const FieldPtrs<6>& eps, const Vec6& E, const Vec6& R,
                     bool add_R, double* partial, double* sumsq6, const XHalo& h, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  hipLaunchKernelGGL(k_eps_norm, dim3(nb), dim3(kBlock), 0, s, g, u, eps, E, R, add_R ? 1 : 0, partial, h,
                     chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_eps_delta(const Grid& g, const FieldPtrs<3>& u, const FieldPtrs<6>& tau, const double* tau_sum, double nvox,
                      const Vec6& E, double coef, const FieldPtrs<6>& eps, double* partial, double* sumsq6, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  hipLaunchKernelGGL(k_eps_delta<0>, dim3(nb), dim3(kBlock), 0, s, g, u, tau, tau_sum, nvox, E, coef, eps, partial,
                     StressParams(), FieldPtrs<kMaxPhases>(), chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_eps_delta_recompute(const Grid& g, const FieldPtrs<3>& u, const FieldPtrs<6>& eps_old, const StressParams& sp,
                                const FieldPtrs<kMaxPhases>& phi, const double* tau_sum, double nvox, const Vec6& E,
                                double coef, const FieldPtrs<6>& eps, double* partial, double* sumsq6, hipStream_t s) {
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  if (sp.pt.n <= 2)
    hipLaunchKernelGGL(k_eps_delta<2>, dim3(nb), dim3(kBlock), 0, s, g, u, eps_old, tau_sum, nvox, E, coef, eps, partial, sp,
                       phi, chunk_rows(g));
  else
    hipLaunchKernelGGL(k_eps_delta<kMaxPhases>, dim3(nb), dim3(kBlock), 0, s, g, u, eps_old, tau_sum, nvox, E, coef, eps,
                       partial, sp, phi, chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_copy(const double* src, double* dst, long ndoubles, hipStream_t s) {
  const long n2 = ndoubles / 2;
  hipLaunchKernelGGL(k_copy, dim3(grid_for(n2, 1 << 16)), dim3(kBlock), 0, s, src, dst, n2);
  FG_HIP_CHECK(hipGetLastError());
}

// Round 4: the same iteration with the CG scalars on the device (a = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small) from
// the sums the sweeps leave there, as in k_cgu_axpy) and the two updates of an iteration in ONE sweep:
//   MODE 5  eps <- eps + a p ;  r <- r - a (p - w)      per-component sums eps_c^2 (partial 0..5) and sum r:r (partial 6)
//   MODE 6  p <- r + a p
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_cg_dev(Grid g, FieldPtrs<6> e, FieldPtrs<6> r, FieldPtrs<6> pp, FieldPtrs<6> w,
                                                   const double* sc, int i_num, int i_den, double nvox, double small,
                                                   double* partial) {
  __shared__ double smem[4 * 7];
  const double a = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small);
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long pidx = (long)blockIdx.x * blockDim.x + threadIdx.x; pidx < npairs; pidx += (long)gridDim.x * blockDim.x) {
    const PairPos p = pair_pos(pidx, g);
    if (p.k >= g.nz) continue;
    const bool second = p.k + 1 < g.nz;
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double2 pv = ld2(pp.p[c], p.off);
      double2 rv = ld2(r.p[c], p.off);
      if (MODE == 5) {
        double2 ev = ld2(e.p[c], p.off);
        const double2 wv = ld2(w.p[c], p.off);
        ev.x = ev.x + a * pv.x;
        ev.y = ev.y + a * pv.y;
        rv.x = rv.x - a * (pv.x - wv.x);
        rv.y = rv.y - a * (pv.y - wv.y);
        st2(e.p[c], p.off, ev);
        st2(r.p[c], p.off, rv);
        acc[c] += ev.x * ev.x + (second ? ev.y * ev.y : 0.0);
        const double wgt = c < 3 ? 1.0 : 2.0;
        sx += wgt * (rv.x * rv.x);
        sy += wgt * (rv.y * rv.y);
      } else {
        st2(pp.p[c], p.off, make_double2(rv.x + a * pv.x, rv.y + a * pv.y));
      }
    }
    if (MODE == 5) acc[6] += sx + (second ? sy : 0.0);
  }
  if (MODE == 5) {
    block_reduce<7>(acc, smem, OpSum());
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 7; ++c) partial[(long)blockIdx.x * 7 + c] = acc[c];
    }
  }
}

void launch_cg_dev(int mode, const Grid& g, const FieldPtrs<6>& e, const FieldPtrs<6>& r, const FieldPtrs<6>& p, const FieldPtrs<6>& w,
                   const double* sc, int i_num, int i_den, double nvox, double small, double* partial, double* out7, hipStream_t s) {
  const int nb = reduce_blocks(g);
  if (mode == 5) {
    hipLaunchKernelGGL((k_cg_dev<5>), dim3(nb), dim3(kBlock), 0, s, g, e, r, p, w, sc, i_num, i_den, nvox, small, partial);
    FG_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, nb, 7, 0.0, out7);
  } else {
    hipLaunchKernelGGL((k_cg_dev<6>), dim3(nb), dim3(kBlock), 0, s, g, e, r, p, w, sc, i_num, i_den, nvox, small, partial);
  }
  FG_HIP_CHECK(hipGetLastError());
}

void launch_cg(int mode, const Grid& g, const FieldPtrs<6>& x, const FieldPtrs<6>& y, const FieldPtrs<6>& z, const Vec6& E,
               double a, double* partial, double* out6, hipStream_t s) {
  const int nb = reduce_blocks(g);
  switch (mode) {
    case 0: hipLaunchKernelGGL((k_cg<0>), dim3(nb), dim3(kBlock), 0, s, g, x, y, z, E, a, partial); break;
    case 1: hipLaunchKernelGGL((k_cg<1>), dim3(nb), dim3(kBlock), 0, s, g, x, y, z, E, a, partial); break;
    case 2: hipLaunchKernelGGL((k_cg<2>), dim3(nb), dim3(kBlock), 0, s, g, x, y, z, E, a, partial); break;
    case 3: hipLaunchKernelGGL((k_cg<3>), dim3(nb), dim3(kBlock), 0, s, g, x, y, z, E, a, partial); break;
    default: hipLaunchKernelGGL((k_cg<4>), dim3(nb), dim3(kBlock), 0, s, g, x, y, z, E, a, partial); break;
  }
  FG_HIP_CHECK(hipGetLastError());
  if (mode != 4) {
    hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, nb, 6, 0.0, out6);
    FG_HIP_CHECK(hipGetLastError());
  }
}

void launch_set_const6(const Grid& g, const FieldPtrs<6>& x, const Vec6& E, hipStream_t s) {
  const long n2 = g.n / 2;
  hipLaunchKernelGGL(k_set_const6, dim3(grid_for(n2, 1 << 16)), dim3(kBlock), 0, s, n2, x, E);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sum6(const Grid& g, const FieldPtrs<6>& x, bool square, double* partial, double* out6, hipStream_t s) {
  const int nb = reduce_blocks(g);
  if (square) hipLaunchKernelGGL((k_sum<6, true>), dim3(nb), dim3(kBlock), 0, s, g, x, partial);
  else hipLaunchKernelGGL((k_sum<6, false>), dim3(nb), dim3(kBlock), 0, s, g, x, partial);
  FG_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, nb, 6, 0.0, out6);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_sum1(const Grid& g, const double* x, double* partial, double* out1, hipStream_t s) {
  const int nb = reduce_blocks(g);
  FieldPtrs<1> f;
  f.p[0] = const_cast<double*>(x);
  hipLaunchKernelGGL((k_sum<1, false>), dim3(nb), dim3(kBlock), 0, s, g, f, partial);
  FG_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, nb, 1, 0.0, out1);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_tangent_minmax(const Grid& g, const PhaseTable& pt, int mixing, const FieldPtrs<kMaxPhases>& phi,
                           double* partial, double* out2, int* error_flag, hipStream_t s) {
  const int nb = reduce_blocks(g);
  hipLaunchKernelGGL(k_tangent_minmax, dim3(nb), dim3(kBlock), 0, s, g, pt, mixing, phi, partial, error_flag);
  FG_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_fold<OpMin>, dim3(1), dim3(kBlock), 0, s, partial, nb, 2, 1.0 / 0.0, out2);
  FG_HIP_CHECK(hipGetLastError());
}

}  // namespace fg
