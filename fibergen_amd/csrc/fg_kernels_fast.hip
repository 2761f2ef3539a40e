// Fast variants of the sweeps of the displacement-based loop: per-voxel effective moduli are
// precomputed (A = sum_p phi_p 2 mu_p, B = sum_p phi_p lambda_p, Voigt mixing F:12752-12761), so the
// polarisation of a voxel is  tau = (A - 2 mu0) eps + (B - lambda0) tr(eps) I  -- two or three flops per
// component instead of the per-phase accumulation, and this translation unit is compiled with FMA
// contraction.  Results agree with the exact-order kernels of fg_kernels.hip to rounding (~1e-16
// relative per operation, asserted to 1e-12 on fields); the exact kernels remain available (u_loop = 1).
#include "fg_kernels.h"

#include <cstdlib>

#include "fg_hip_util.h"
#include "fg_kernels_common.h"

namespace fg {

namespace {

// Position of a thread's z pair for the displacement sweep.
struct UPos {
  int i, j, k;          // voxel (i, j, k) and (i, j, k+1)
  long ro;              // element offset of row (i, j)
  bool second;          // the pair holds two voxels (false only for the last pair of an odd nz)
  bool prev_ok, next_ok;  // lanes -1 / +1 of the wave hold the adjacent pairs of the same row
};

// u_k -> strain of the two voxels of a pair (eout) and the divergence of their polarisation (fout).
// The sweep holds, per (x,y) row it touches, the vector [z = k-1, k, k+1, k+2]: the pair (k, k+1) is one
// 16-byte load; the outer two values are the neighbouring lanes' pair halves, fetched by a wave shift.
// Only the lanes where prev_ok / next_ok is false load them from memory (edge loads, grouped so that
// they are issued together).  Must be called by all 64 lanes of a wave.
// ODD: nz is odd, the last pair of a row holds one voxel and z+1 wraps to 0.
template <bool ODD>
__device__ __forceinline__ void u_fast_pair(const Grid& g, double beta, double gamma, const FieldPtrs<3>& u,
                                            const FieldPtrs<2>& mod, const Vec6& E, const UPos& p, double (&fout)[2][3],
                                            double (&eout)[2][6]) {
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const bool second = p.second, prev_ok = p.prev_ok, next_ok = p.next_ok;
  const long xf = (p.i + 1 == g.nx ? -(long)(g.nx - 1) : 1L) * g.nyzp;
  const long xb = (p.i == 0 ? (long)(g.nx - 1) : -1L) * g.nyzp;
  const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
  const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
  const long ro = p.ro;
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
#define FG_ROW(name, a, off)                         \
  Row4 name;                                         \
  {                                                  \
    const double2 d_ = ld2(a, rk + (off));           \
    name.v[1] = d_.x;                                \
    name.v[2] = d_.y;                                \
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
}

template <bool ODD>
__global__ __launch_bounds__(kBlock) void k_u_fast(Grid g, double beta, double gamma, FieldPtrs<3> u, FieldPtrs<2> mod,
                                                   FieldPtrs<3> fo, Vec6 E, double* partial, Sweep sw) {
  __shared__ double smem[4 * 6];
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx_raw = (run.first + it * run.stride) * kBlock + threadIdx.x;
    const long pidx = pidx_raw < npairs ? pidx_raw : npairs - 1;  // clamped lanes discard their result
    const PairPos pp = pair_pos_tiled(pidx, g, sw);
    const bool valid = pidx_raw < npairs && pp.k < g.nz;
    const int lane = threadIdx.x & 63;
    UPos p;
    p.i = pp.i; p.j = pp.j; p.k = pp.k;
    p.ro = pp.off - pp.k;
    p.second = !ODD || pp.k + 1 < g.nz;
    p.prev_ok = lane > 0 && pp.k > 0;
    p.next_ok = lane < 63 && pp.k + 2 < g.nz && pidx_raw + 1 < npairs;
    double fout[2][3], eout[2][6];
    u_fast_pair<ODD>(g, beta, gamma, u, mod, E, p, fout, eout);
    if (valid) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double ey = p.second ? eout[1][c] : 0.0;
        acc[c] += eout[0][c] * eout[0][c] + ey * ey;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) st2(fo.p[c], pp.off, make_double2(fout[0][c], p.second ? fout[1][c] : 0.0));
    }
  }
  block_reduce<6>(acc, smem, OpSum());
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) partial[(long)blockIdx.x * 6 + c] = acc[c];
  }
}

// ------------------------------------------------------------------------------------------------
// Tiled variant of the displacement sweep: every strain and every polarisation component is computed ONCE.
//
// k_u_fast recomputes the strain of the six face neighbours in registers (3.5x redundant arithmetic, 25 row
// loads per voxel pair) and is bound by VALU issue.  Here a workgroup of TYR waves owns a brick: wave r is the
// y row j0-1+r of the tile, its 64 lanes are consecutive z pairs, and the workgroup marches along x.
//   * x neighbours are the thread's own values of the previous / next plane (registers);
//   * y neighbours (u and tau of rows r-1, r+1) go through LDS, one exchange each per plane;
//   * z neighbours are the adjacent lanes (DPP wave shifts).
// Row 0, row TYR-1, lane 0 and lane 63 are halo: they compute their strain and polarisation like everyone
// else but own no output, so no lane needs a value from outside the workgroup (no edge loads, no recomputation).
// ZS >= 1: nz/2 == 64 ZS, a z row is ZS whole waves of the workgroup (no halo lanes; the first / last lane of a wave takes
// its z neighbour from the adjacent wave's entry in LDS, periodically).  ZS == 0: general nz, lanes 0 and 63 are halo.
// Per plane and thread: 5 16-byte global loads (u x3, A, B), 6 LDS writes + 6 LDS reads, 2 barriers.
// f0 of plane q is complete at step q; f1, f2 need tau5, tau4 of plane q+1 and are finished one step later.
// tools/utile_probe.hip compiles this file with -DFG_PROBE_K1: cycle stamps (s_memtime) of one wave per sampled
// workgroup inside one marching step of the tiled sweep.  Empty in the library.
#ifdef FG_PROBE_K1
constexpr int kK1ProbeBlocks = 64, kK1ProbeSlots = 16;
__device__ unsigned long long g_k1_probe[kK1ProbeBlocks][kK1ProbeSlots];
#define FG_K1_MARK(slot)                                                                                   \
  do {                                                                                                     \
    if ((st == FG_PROBE_K1_STEP || (slot) == 0 || (slot) >= 8) && threadIdx.x == FG_PROBE_K1_THREAD &&     \
        blockIdx.x % FG_PROBE_K1_STRIDE == 0 && blockIdx.x / FG_PROBE_K1_STRIDE < kK1ProbeBlocks)           \
      g_k1_probe[blockIdx.x / FG_PROBE_K1_STRIDE][(slot)] = __builtin_readcyclecounter();                  \
  } while (0)
#else
#define FG_K1_MARK(slot) do { } while (0)
#endif

// PHI2: two complementary phases (phi_0 = 1 - phi_1 everywhere, checked by the caller): the sweep reads phi_1 (mod.p[0]) and
// forms the effective moduli itself, A = 2 mu_0 + phi_1 (2 mu_1 - 2 mu_0), B likewise -- 8 bytes per voxel less than the two
// precomputed arrays (64 -> 56 B/voxel, the algorithmic figure).  lin = {2 mu_0 + beta, 2 (mu_1 - mu_0), lambda_0 + gamma,
// lambda_1 - lambda_0} of the two phases.
struct PhaseLin {
  double a0, da, b0, db;
};

// CGP: the sweep's displacement is the NEW search direction of the conjugate gradients, formed on the fly:
// u := r + b u  (u_p = u_r + beta u_p, F:23235-23240) with b = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small) from the
// sums the dot sweeps left on the device, stored to po wherever this thread owns the voxels (out of place: halo rows and
// lanes form the value of voxels the neighbouring tile owns from the OLD direction) -- the separate update sweep
// (k_cgu_axpy<1>: 72 B per voxel) becomes 24 B more read and 24 B more written here.
struct CgDirection {
  const double* r[3];
  double* po[3];
  const double* sc;
  int i_num, i_den;
  double nvox, small;
};

template <int TYR, int ZS, bool SUMT, bool PHI2, bool CGP = false>
__global__ __launch_bounds__(TYR * (ZS ? ZS : 1) * 64) void k_u_tile(Grid g, double beta, double gamma, FieldPtrs<3> u,
                                                                      FieldPtrs<2> mod, FieldPtrs<3> fo, Vec6 E, double* partial,
                                                                      int nty, int ntz, int LX, int nt, PhaseLin lin,
                                                                      CgDirection cg) {
  constexpr bool FULLROW = ZS > 0;
  constexpr int NZS = ZS ? ZS : 1;        // waves per row
  constexpr int TYU = TYR - 2;            // rows with output
  constexpr int TZU = FULLROW ? 64 * NZS : 62;  // pairs with output per tile row
  constexpr int RW = NZS * 64;            // LDS entries per row
  extern __shared__ __align__(16) double2 tile_lds[];
  double2(*Ub)[TYR][RW] = reinterpret_cast<double2(*)[TYR][RW]>(tile_lds);                  // [3][TYR][RW]
  double2(*Tb)[TYR][RW] = reinterpret_cast<double2(*)[TYR][RW]>(tile_lds + 3 * TYR * RW);   // [3][TYR][RW]
  // SUMT: also the six sums of the polarisation (<tau> drives the mixed boundary conditions): 12 values per workgroup
  constexpr int NS = SUMT ? 12 : 6;
  __shared__ double red[TYR * NZS * NS];
  __shared__ double edge[3][TYR][NZS];    // tau2.y of lane 63, tau3.x and tau4.x of lane 0 of every wave (ZS >= 1)

  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r = wv / NZS, zs = wv % NZS;  // tile row, z segment of the row
  const int li = zs * 64 + l;             // entry of this thread in an LDS row
  const int zprev = (zs + NZS - 1) % NZS, znext = (zs + 1) % NZS;
  const int nzh = g.nz / 2;
  // workgroups of one XCD (blockIdx % 8) take a contiguous run of tiles: z- and y-adjacent tiles, which share halo
  // rows and 128-byte segments, then meet in that XCD's L2 (FETCH_SIZE 2.1x -> see profiles/ for the effect)
  int b = blockIdx.x;
  {
    const int nb = gridDim.x;
    if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
  }
  const int tz = b % ntz;
  b /= ntz;
  const int ty = b % nty;
  const int tx = b / nty;
  const bool surplus = tx * LX >= g.nx;   // padding workgroup (grid rounded up to a multiple of 8)
  const int j0 = min(ty * TYU, g.ny - TYU), kp0 = min(tz * TZU, nzh - TZU), x0 = surplus ? 0 : tx * LX;
  const int jr = j0 - 1 + r;                              // may be -1 or ny
  const int j = jr < 0 ? jr + g.ny : (jr >= g.ny ? jr - g.ny : jr);
  const int kr = FULLROW ? li : kp0 - 1 + l;
  const int kp = kr < 0 ? kr + nzh : (kr >= nzh ? kr - nzh : kr);
  const bool own = r >= 1 && r <= TYU && jr >= ty * TYU && (FULLROW || (l >= 1 && l <= TZU && kr >= tz * TZU));
  const long rowoff = (long)j * g.nzp + 2 * kp;
  const int rm = r > 0 ? r - 1 : 0, rp = r + 1 < TYR ? r + 1 : TYR - 1;
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const int nsteps = surplus ? -2 : (x0 + LX <= g.nx ? LX : g.nx - x0);   // surplus: no steps at all

  auto plane = [&](int q) {  // element offset of x plane q (periodic, or the halo planes of an x-slab; q in [-1, 2 nx))
    const int x = q < 0 ? q + g.xw_lo : (q >= g.nx ? q - g.xw_hi : q);
    return (long)x * g.nyzp + rowoff;
  };
  // f is read back by the z pass only after the whole sweep: with fields larger than the Infinity Cache it is stored
  // past the cache (nt), see cstore_stream
  auto store_f = [&](double* base, long off, double2 v) {
    if (nt) {
      typedef double fg_v2d __attribute__((ext_vector_type(2)));
      fg_v2d t;
      t.x = v.x;
      t.y = v.y;
      __builtin_nontemporal_store(t, reinterpret_cast<fg_v2d*>(base + off));
    } else {
      st2(base, off, v);
    }
  };
  auto prev_y = [&](double v) { return dpp_move<0x138>(v); };  // lane i <- i-1 (lane 0: fixed up below when ZS >= 1)
  auto next_x = [&](double v) { return dpp_move<0x130>(v); };  // lane i <- i+1 (lane 63: likewise)

  const double cgb = CGP ? (cg.sc[cg.i_num] / cg.nvox + cg.small) / (cg.sc[cg.i_den] / cg.nvox + cg.small) : 0.0;
  // the displacement of component c at offset o (CGP: the new direction, kept where `keep` says the plane is this tile's)
  auto load_u = [&](int c, long o, bool keep) {
    double2 v = ld2(u.p[c], o);
    if (CGP) {
      const double2 rv = ld2(cg.r[c], o);
      v.x = rv.x + cgb * v.x;
      v.y = rv.y + cgb * v.y;
      if (keep && own) st2(cg.po[c], o, v);
    }
    return v;
  };
  double2 uc[3], un[3], u2[3];
  {
    const long o0 = plane(x0 - 1), o1 = plane(x0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uc[c] = load_u(c, o0, false);
      un[c] = load_u(c, o1, nsteps > 0);
    }
  }
  double2 dx1 = make_double2(0.0, 0.0), dx2 = dx1;          // U1, U2 minus their previous plane (warm-up: unused)
  double2 t0m = dx1, t5m = dx1, t4m = dx1, part1 = dx1, part2 = dx1;
  double acc[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) acc[c] = 0.0;
  {
    [[maybe_unused]] const int st = -100;
    FG_K1_MARK(0);   // workgroup start (after the first loads were issued)
  }

  for (int st = -1; st <= nsteps; ++st) {
    const int q = x0 + st;                                   // plane of this step
    FG_K1_MARK(1);
    // u two planes ahead (consumed next step); the moduli of this plane are first needed after the LDS exchange
    const long o2 = plane(q + 2), oq = plane(q);
    // (the moduli first: they are this step's critical load -- loads return in order (vmcnt), and behind the three prefetches of
    // plane q + 2 the wait for them was `s_waitcnt vmcnt(0)`, now vmcnt(3).  r5, one job, old / new library alternating: 128^3
    // 7 666-7 690 -> 7 736-7 743 it/s, 256^3 and 512^3 within the noise -- the steps wait on the memory queue, not on one load)
    const double2 Ac = ld2(mod.p[0], oq);
    double2 Bc = Ac;
    if (!PHI2) Bc = ld2(mod.p[1], oq);
#pragma unroll
    for (int c = 0; c < 3; ++c) u2[c] = load_u(c, o2, st + 2 >= 0 && st + 2 < nsteps);
    // ---- y neighbours of u through LDS
#pragma unroll
    for (int c = 0; c < 3; ++c) Ub[c][r][li] = uc[c];
    FG_K1_MARK(2);
    __syncthreads();
    FG_K1_MARK(3);
    const double2 U0yb = Ub[0][rm][li], U1yf = Ub[1][rp][li], U2yb = Ub[2][rm][li];
    double U0zb = prev_y(uc[0].y), U1zb = prev_y(uc[1].y), U2zf = next_x(uc[2].x);
    if (FULLROW) {   // wave edges: the neighbour pair lives in the adjacent wave of the same row
      if (l == 0) {
        U0zb = Ub[0][r][zprev * 64 + 63].y;
        U1zb = Ub[1][r][zprev * 64 + 63].y;
      }
      if (l == 63) U2zf = Ub[2][r][znext * 64].x;
    }
    // ---- strain of the two voxels  (F:18632-18686)
    double2 e0, e1, e2, e3, e4, e5;
    e0.x = E.v[0] + (un[0].x - uc[0].x) * hx;
    e0.y = E.v[0] + (un[0].y - uc[0].y) * hx;
    e1.x = E.v[1] + (U1yf.x - uc[1].x) * hy;
    e1.y = E.v[1] + (U1yf.y - uc[1].y) * hy;
    e2.x = E.v[2] + (uc[2].y - uc[2].x) * hz;
    e2.y = E.v[2] + (U2zf - uc[2].y) * hz;
    e3.x = E.v[3] + 0.5 * ((uc[2].x - U2yb.x) * hy + (uc[1].x - U1zb) * hz);
    e3.y = E.v[3] + 0.5 * ((uc[2].y - U2yb.y) * hy + (uc[1].y - uc[1].x) * hz);
    e4.x = E.v[4] + 0.5 * (dx2.x * hx + (uc[0].x - U0zb) * hz);
    e4.y = E.v[4] + 0.5 * (dx2.y * hx + (uc[0].y - uc[0].x) * hz);
    e5.x = E.v[5] + 0.5 * (dx1.x * hx + (uc[0].x - U0yb.x) * hy);
    e5.y = E.v[5] + 0.5 * (dx1.y * hx + (uc[0].y - U0yb.y) * hy);
    // ---- polarisation  tau = (A - 2 mu0) eps + (B - lambda0) tr(eps) I
    const double ax = PHI2 ? lin.a0 + Ac.x * lin.da : Ac.x + beta, ay = PHI2 ? lin.a0 + Ac.y * lin.da : Ac.y + beta;
    const double bx = PHI2 ? lin.b0 + Bc.x * lin.db : Bc.x + gamma, by = PHI2 ? lin.b0 + Bc.y * lin.db : Bc.y + gamma;
    const double trx = e0.x + e1.x + e2.x, try_ = e0.y + e1.y + e2.y;
    double2 t0, t1, t2, t3, t4, t5;
    t0.x = e0.x * ax + bx * trx; t0.y = e0.y * ay + by * try_;
    t1.x = e1.x * ax + bx * trx; t1.y = e1.y * ay + by * try_;
    t2.x = e2.x * ax + bx * trx; t2.y = e2.y * ay + by * try_;
    t3.x = e3.x * ax; t3.y = e3.y * ay;
    t4.x = e4.x * ax; t4.y = e4.y * ay;
    t5.x = e5.x * ax; t5.y = e5.y * ay;
    const bool inside = st >= 0 && st < nsteps;
    if (own && inside) {
      acc[0] += e0.x * e0.x + e0.y * e0.y; acc[1] += e1.x * e1.x + e1.y * e1.y; acc[2] += e2.x * e2.x + e2.y * e2.y;
      acc[3] += e3.x * e3.x + e3.y * e3.y; acc[4] += e4.x * e4.x + e4.y * e4.y; acc[5] += e5.x * e5.x + e5.y * e5.y;
      if (SUMT) {
        acc[NS - 6] += t0.x + t0.y; acc[NS - 5] += t1.x + t1.y; acc[NS - 4] += t2.x + t2.y;
        acc[NS - 3] += t3.x + t3.y; acc[NS - 2] += t4.x + t4.y; acc[NS - 1] += t5.x + t5.y;
      }
    }
    FG_K1_MARK(4);
    // ---- y neighbours of tau through LDS
    Tb[0][r][li] = t1;
    Tb[1][r][li] = t5;
    Tb[2][r][li] = t3;
    if (FULLROW) {
      if (l == 63) edge[0][r][zs] = t2.y;
      if (l == 0) {
        edge[1][r][zs] = t3.x;
        edge[2][r][zs] = t4.x;
      }
    }
    FG_K1_MARK(5);
    __syncthreads();
    FG_K1_MARK(6);
    const double2 t1yb = Tb[0][rm][li], t5yf = Tb[1][rp][li], t3yf = Tb[2][rp][li];
    double t2zb = prev_y(t2.y), t3zf = next_x(t3.x), t4zf = next_x(t4.x);
    if (FULLROW) {
      if (l == 0) t2zb = edge[0][r][zprev];
      if (l == 63) {
        t3zf = edge[1][r][znext];
        t4zf = edge[2][r][znext];
      }
    }
    // ---- divergence: f0 of this plane, f1 / f2 of the previous one
    if (own) {
      if (inside) {
        double2 f0;
        f0.x = (t0.x - t0m.x) * hx + (t5yf.x - t5.x) * hy + (t4.y - t4.x) * hz;
        f0.y = (t0.y - t0m.y) * hx + (t5yf.y - t5.y) * hy + (t4zf - t4.y) * hz;
        store_f(fo.p[0], oq, f0);
      }
      if (st >= 1) {
        const long op = plane(q - 1);
        store_f(fo.p[1], op, make_double2((t5.x - t5m.x) * hx + part1.x, (t5.y - t5m.y) * hx + part1.y));
        store_f(fo.p[2], op, make_double2((t4.x - t4m.x) * hx + part2.x, (t4.y - t4m.y) * hx + part2.y));
      }
    }
    part1.x = (t1.x - t1yb.x) * hy + (t3.y - t3.x) * hz;
    part1.y = (t1.y - t1yb.y) * hy + (t3zf - t3.y) * hz;
    part2.x = (t3yf.x - t3.x) * hy + (t2.x - t2zb) * hz;
    part2.y = (t3yf.y - t3.y) * hy + (t2.y - t2.x) * hz;
    // ---- advance one plane
    t0m = t0; t5m = t5; t4m = t4;
    dx1.x = un[1].x - uc[1].x; dx1.y = un[1].y - uc[1].y;
    dx2.x = un[2].x - uc[2].x; dx2.y = un[2].y - uc[2].y;
#pragma unroll
    for (int c = 0; c < 3; ++c) { uc[c] = un[c]; un[c] = u2[c]; }
    FG_K1_MARK(7);
  }
  {
    [[maybe_unused]] const int st = -100;
    FG_K1_MARK(8);   // end of the march
  }
  // ---- sums of squares: fixed-order reduction over the workgroup
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    double a = acc[c];
    a += dpp_move<0x128>(a);
    a += dpp_move<0x124>(a);
    a += dpp_move<0x122>(a);
    a += dpp_move<0x121>(a);
    acc[c] = (read_lane(a, 0) + read_lane(a, 16)) + (read_lane(a, 32) + read_lane(a, 48));
  }
  if (l == 0) {
#pragma unroll
    for (int c = 0; c < NS; ++c) red[wv * NS + c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double a = 0.0;
    for (int w = 0; w < TYR * NZS; ++w) a += red[w * NS + threadIdx.x];
    partial[(long)blockIdx.x * NS + threadIdx.x] = a;
  }
}

// Tiled marching form of the vector sweeps of the conjugate gradients in displacement space (k_cgu_dot / k_cgu_axpy,
// fg_kernels.hip; innerProductL2 F:20955-21038, the updates of runCGElasticity F:23201-23240): k_u_tile's tiling (rows = waves,
// lanes = z pairs, march along x; x neighbours in registers, y neighbours through LDS, z neighbours by DPP) for the staggered
// gradients of TWO displacement fields A and B at once, every value loaded once per tile.
//   MODE 0:  A = a, B = b as stored;  partial[0] = sum grad_s A : (grad_s A - grad_s B)               (p : (p - w))
//   MODE 1:  A = a + alpha y,  B = b - alpha (y - w)  (eps += alpha p ; r -= alpha (p - w) with the alpha the device holds),
//            stored to ao / bo -- OUT OF PLACE: the halo rows of a tile re-evaluate the update of rows the neighbouring tile owns,
//            which must still find the old values --, partial[0..5] = sums of (E + grad_s A)_c^2, partial[6] = sum grad_s B : grad_s B.
// The untiled pair (axpy sweep 144 B per voxel + dot sweep 48 B with recomputed neighbours) took 0.46 + 0.23 ms at 256^3.
template <int TYR, int ZS, int MODE>
__global__ __launch_bounds__(TYR * (ZS ? ZS : 1) * 64) void k_cgu_tile(Grid g, FieldPtrs<3> a, FieldPtrs<3> b, FieldPtrs<3> y,
                                                                        FieldPtrs<3> w, FieldPtrs<3> ao, FieldPtrs<3> bo, Vec6 E,
                                                                        const double* sc, int i_num, int i_den, double nvox,
                                                                        double small, double* partial, int nty, int ntz, int LX,
                                                                        int nt) {
  constexpr bool FULLROW = ZS > 0;
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2;
  constexpr int TZU = FULLROW ? 64 * NZS : 62;
  constexpr int RW = NZS * 64;
  constexpr int NS = 7;
  extern __shared__ __align__(16) double2 tile_lds[];
  double2(*Xb)[TYR][RW] = reinterpret_cast<double2(*)[TYR][RW]>(tile_lds);   // [6][TYR][RW]: A0 A1 A2 B0 B1 B2 of the plane
  __shared__ double red[TYR * NZS * NS];

  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r = wv / NZS, zs = wv % NZS;
  const int li = zs * 64 + l;
  const int zprev = (zs + NZS - 1) % NZS, znext = (zs + 1) % NZS;
  const int nzh = g.nz / 2;
  int bi = blockIdx.x;
  {
    const int nb = gridDim.x;
    if (nb % 8 == 0) bi = (bi % 8) * (nb / 8) + bi / 8;
  }
  const int tz = bi % ntz;
  bi /= ntz;
  const int ty = bi % nty;
  const int tx = bi / nty;
  const bool surplus = tx * LX >= g.nx;
  const int j0 = min(ty * TYU, g.ny - TYU), kp0 = min(tz * TZU, nzh - TZU), x0 = surplus ? 0 : tx * LX;
  const int jr = j0 - 1 + r;
  const int j = jr < 0 ? jr + g.ny : (jr >= g.ny ? jr - g.ny : jr);
  const int kr = FULLROW ? li : kp0 - 1 + l;
  const int kp = kr < 0 ? kr + nzh : (kr >= nzh ? kr - nzh : kr);
  const bool own = r >= 1 && r <= TYU && jr >= ty * TYU && (FULLROW || (l >= 1 && l <= TZU && kr >= tz * TZU));
  const long rowoff = (long)j * g.nzp + 2 * kp;
  const int rm = r > 0 ? r - 1 : 0, rp = r + 1 < TYR ? r + 1 : TYR - 1;
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const int nsteps = surplus ? -2 : (x0 + LX <= g.nx ? LX : g.nx - x0);
  const double al = MODE == 1 ? (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small) : 0.0;

  auto plane = [&](int q) {
    const int x = q < 0 ? q + g.xw_lo : (q >= g.nx ? q - g.xw_hi : q);
    return (long)x * g.nyzp + rowoff;
  };
  auto store = [&](double* base, long off, double2 v) {
    if (nt) {
      typedef double fg_v2d __attribute__((ext_vector_type(2)));
      fg_v2d t;
      t.x = v.x;
      t.y = v.y;
      __builtin_nontemporal_store(t, reinterpret_cast<fg_v2d*>(base + off));
    } else {
      st2(base, off, v);
    }
  };
  auto prev_y = [&](double v) { return dpp_move<0x138>(v); };
  auto next_x = [&](double v) { return dpp_move<0x130>(v); };
  // the two fields of plane q (as stored, or updated), and their store where this thread owns the voxels of an owned plane
  auto fetch = [&](int q, double2 (&A)[3], double2 (&B)[3], bool keep) {
    const long o = plane(q);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      A[c] = ld2(a.p[c], o);
      B[c] = ld2(b.p[c], o);
      if (MODE == 1) {
        const double2 yv = ld2(y.p[c], o), wv2 = ld2(w.p[c], o);
        A[c].x = A[c].x + al * yv.x;
        A[c].y = A[c].y + al * yv.y;
        B[c].x = B[c].x - al * (yv.x - wv2.x);
        B[c].y = B[c].y - al * (yv.y - wv2.y);
        if (keep && own) {
          store(ao.p[c], o, A[c]);
          store(bo.p[c], o, B[c]);
        }
      }
    }
  };

  double acc[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) acc[c] = 0.0;
  if (nsteps > 0) {   // uniform per workgroup
    double2 Ac[3], Bc[3], An[3], Bn[3];
    fetch(x0 - 1, Ac, Bc, false);
    fetch(x0, An, Bn, true);
    double2 dA1 = make_double2(0.0, 0.0), dA2 = dA1, dB1 = dA1, dB2 = dA1;
    for (int st = -1; st < nsteps; ++st) {
      const int q = x0 + st;
      // y neighbours through LDS
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Xb[c][r][li] = Ac[c];
        Xb[3 + c][r][li] = Bc[c];
      }
      __syncthreads();
      const double2 A0yb = Xb[0][rm][li], A1yf = Xb[1][rp][li], A2yb = Xb[2][rm][li];
      const double2 B0yb = Xb[3][rm][li], B1yf = Xb[4][rp][li], B2yb = Xb[5][rm][li];
      double A0zb = prev_y(Ac[0].y), A1zb = prev_y(Ac[1].y), A2zf = next_x(Ac[2].x);
      double B0zb = prev_y(Bc[0].y), B1zb = prev_y(Bc[1].y), B2zf = next_x(Bc[2].x);
      if (FULLROW) {
        if (l == 0) {
          A0zb = Xb[0][r][zprev * 64 + 63].y;
          A1zb = Xb[1][r][zprev * 64 + 63].y;
          B0zb = Xb[3][r][zprev * 64 + 63].y;
          B1zb = Xb[4][r][zprev * 64 + 63].y;
        }
        if (l == 63) {
          A2zf = Xb[2][r][znext * 64].x;
          B2zf = Xb[5][r][znext * 64].x;
        }
      }
      __syncthreads();   // the image is free for the next plane
      if (own && st >= 0) {
        // staggered symmetric gradients of the two voxels  (F:18632-18686 without the prescribed strain)
        double2 ga[6], gb[6];
        ga[0].x = (An[0].x - Ac[0].x) * hx;  ga[0].y = (An[0].y - Ac[0].y) * hx;
        ga[1].x = (A1yf.x - Ac[1].x) * hy;   ga[1].y = (A1yf.y - Ac[1].y) * hy;
        ga[2].x = (Ac[2].y - Ac[2].x) * hz;  ga[2].y = (A2zf - Ac[2].y) * hz;
        ga[3].x = 0.5 * ((Ac[2].x - A2yb.x) * hy + (Ac[1].x - A1zb) * hz);
        ga[3].y = 0.5 * ((Ac[2].y - A2yb.y) * hy + (Ac[1].y - Ac[1].x) * hz);
        ga[4].x = 0.5 * (dA2.x * hx + (Ac[0].x - A0zb) * hz);
        ga[4].y = 0.5 * (dA2.y * hx + (Ac[0].y - Ac[0].x) * hz);
        ga[5].x = 0.5 * (dA1.x * hx + (Ac[0].x - A0yb.x) * hy);
        ga[5].y = 0.5 * (dA1.y * hx + (Ac[0].y - A0yb.y) * hy);
        gb[0].x = (Bn[0].x - Bc[0].x) * hx;  gb[0].y = (Bn[0].y - Bc[0].y) * hx;
        gb[1].x = (B1yf.x - Bc[1].x) * hy;   gb[1].y = (B1yf.y - Bc[1].y) * hy;
        gb[2].x = (Bc[2].y - Bc[2].x) * hz;  gb[2].y = (B2zf - Bc[2].y) * hz;
        gb[3].x = 0.5 * ((Bc[2].x - B2yb.x) * hy + (Bc[1].x - B1zb) * hz);
        gb[3].y = 0.5 * ((Bc[2].y - B2yb.y) * hy + (Bc[1].y - Bc[1].x) * hz);
        gb[4].x = 0.5 * (dB2.x * hx + (Bc[0].x - B0zb) * hz);
        gb[4].y = 0.5 * (dB2.y * hx + (Bc[0].y - Bc[0].x) * hz);
        gb[5].x = 0.5 * (dB1.x * hx + (Bc[0].x - B0yb.x) * hy);
        gb[5].y = 0.5 * (dB1.y * hx + (Bc[0].y - B0yb.y) * hy);
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double wgt = c < 3 ? 1.0 : 2.0;
          if (MODE == 0) {
            sx += wgt * (ga[c].x * (ga[c].x - gb[c].x));
            sy += wgt * (ga[c].y * (ga[c].y - gb[c].y));
          } else {
            const double ex = E.v[c] + ga[c].x, ey = E.v[c] + ga[c].y;
            acc[c] += ex * ex + ey * ey;
            sx += wgt * (gb[c].x * gb[c].x);
            sy += wgt * (gb[c].y * gb[c].y);
          }
        }
        acc[MODE == 0 ? 0 : 6] += sx + sy;
      }
      // advance one plane
      dA1.x = An[1].x - Ac[1].x; dA1.y = An[1].y - Ac[1].y;
      dA2.x = An[2].x - Ac[2].x; dA2.y = An[2].y - Ac[2].y;
      dB1.x = Bn[1].x - Bc[1].x; dB1.y = Bn[1].y - Bc[1].y;
      dB2.x = Bn[2].x - Bc[2].x; dB2.y = Bn[2].y - Bc[2].y;
#pragma unroll
      for (int c = 0; c < 3; ++c) { Ac[c] = An[c]; Bc[c] = Bn[c]; }
      if (st + 1 < nsteps) fetch(q + 2, An, Bn, st + 2 < nsteps);   // plane q + 2: the forward x neighbour of the next step
    }
  }
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    double v = acc[c];
    v += dpp_move<0x128>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x122>(v);
    v += dpp_move<0x121>(v);
    acc[c] = (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
  }
  if (l == 0) {
#pragma unroll
    for (int c = 0; c < NS; ++c) red[wv * NS + c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double v = 0.0;
    for (int q = 0; q < TYR * NZS; ++q) v += red[q * NS + threadIdx.x];
    partial[(long)blockIdx.x * NS + threadIdx.x] = v;
  }
}

// The second half of k_u_tile on its own, for passes whose state is the strain field (viscosity mode; mixed boundary
// conditions): eps -> f = div((C - C0) : eps) and the six sums of the polarisation, Voigt mixing with the precomputed
// effective moduli.  Same tiling (rows = waves, lanes = z pairs, march along x, y neighbours of tau through LDS -- two
// images, one barrier per plane --, z neighbours by DPP); the strain of the next plane is requested one step ahead.
template <int TYR, int ZS>
__global__ __launch_bounds__(TYR * (ZS ? ZS : 1) * 64) void k_eps_tile(Grid g, double beta, double gamma, FieldPtrs<6> eps,
                                                                        FieldPtrs<2> mod, FieldPtrs<3> fo, double* partial,
                                                                        int nty, int ntz, int LX, int nt) {
  constexpr bool FULLROW = ZS > 0;
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2;
  constexpr int TZU = FULLROW ? 64 * NZS : 62;
  constexpr int RW = NZS * 64;
  extern __shared__ __align__(16) double2 tile_lds[];
  double2(*Tb)[3][TYR][RW] = reinterpret_cast<double2(*)[3][TYR][RW]>(tile_lds);   // [2][3][TYR][RW]
  __shared__ double red[TYR * NZS * 6];
  __shared__ double edge[2][3][TYR][NZS];

  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r = wv / NZS, zs = wv % NZS;
  const int li = zs * 64 + l;
  const int zprev = (zs + NZS - 1) % NZS, znext = (zs + 1) % NZS;
  const int nzh = g.nz / 2;
  int b = blockIdx.x;
  {
    const int nb = gridDim.x;
    if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
  }
  const int tz = b % ntz;
  b /= ntz;
  const int ty = b % nty;
  const int tx = b / nty;
  const bool surplus = tx * LX >= g.nx;
  const int j0 = min(ty * TYU, g.ny - TYU), kp0 = min(tz * TZU, nzh - TZU), x0 = surplus ? 0 : tx * LX;
  const int jr = j0 - 1 + r;
  const int j = jr < 0 ? jr + g.ny : (jr >= g.ny ? jr - g.ny : jr);
  const int kr = FULLROW ? li : kp0 - 1 + l;
  const int kp = kr < 0 ? kr + nzh : (kr >= nzh ? kr - nzh : kr);
  const bool own = r >= 1 && r <= TYU && jr >= ty * TYU && (FULLROW || (l >= 1 && l <= TZU && kr >= tz * TZU));
  const long rowoff = (long)j * g.nzp + 2 * kp;
  const int rm = r > 0 ? r - 1 : 0, rp = r + 1 < TYR ? r + 1 : TYR - 1;
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const int nsteps = surplus ? -2 : (x0 + LX <= g.nx ? LX : g.nx - x0);

  auto plane = [&](int q) {
    const int x = q < 0 ? q + g.nx : (q >= g.nx ? q - g.nx : q);
    return (long)x * g.nyzp + rowoff;
  };
  auto store_f = [&](double* base, long off, double2 v) {
    if (nt) {
      typedef double fg_v2d __attribute__((ext_vector_type(2)));
      fg_v2d t;
      t.x = v.x;
      t.y = v.y;
      __builtin_nontemporal_store(t, reinterpret_cast<fg_v2d*>(base + off));
    } else {
      st2(base, off, v);
    }
  };
  auto prev_y = [&](double v) { return dpp_move<0x138>(v); };
  auto next_x = [&](double v) { return dpp_move<0x130>(v); };

  double2 en[6], An, Bn;   // inputs of the coming step
  {
    const long o = plane(x0 - 1);
#pragma unroll
    for (int c = 0; c < 6; ++c) en[c] = ld2(eps.p[c], o);
    An = ld2(mod.p[0], o);
    Bn = ld2(mod.p[1], o);
  }
  double2 zero = make_double2(0.0, 0.0);
  double2 t0m = zero, t5m = zero, t4m = zero, part1 = zero, part2 = zero;
  double acc[6] = {0, 0, 0, 0, 0, 0};

  for (int st = -1; st <= nsteps; ++st) {
    const int q = x0 + st;
    const long oq = plane(q);
    double2 e0 = en[0], e1 = en[1], e2 = en[2], e3 = en[3], e4 = en[4], e5 = en[5];
    const double2 Ac = An, Bc = Bn;
    if (st < nsteps) {
      const long on = plane(q + 1);
#pragma unroll
      for (int c = 0; c < 6; ++c) en[c] = ld2(eps.p[c], on);
      An = ld2(mod.p[0], on);
      Bn = ld2(mod.p[1], on);
    }
    // ---- polarisation  tau = (A - 2 mu0) eps + (B - lambda0) tr(eps) I
    const double ax = Ac.x + beta, ay = Ac.y + beta, bx = Bc.x + gamma, by = Bc.y + gamma;
    const double trx = e0.x + e1.x + e2.x, try_ = e0.y + e1.y + e2.y;
    double2 t0, t1, t2, t3, t4, t5;
    t0.x = e0.x * ax + bx * trx; t0.y = e0.y * ay + by * try_;
    t1.x = e1.x * ax + bx * trx; t1.y = e1.y * ay + by * try_;
    t2.x = e2.x * ax + bx * trx; t2.y = e2.y * ay + by * try_;
    t3.x = e3.x * ax; t3.y = e3.y * ay;
    t4.x = e4.x * ax; t4.y = e4.y * ay;
    t5.x = e5.x * ax; t5.y = e5.y * ay;
    const bool inside = st >= 0 && st < nsteps;
    if (own && inside) {
      acc[0] += t0.x + t0.y; acc[1] += t1.x + t1.y; acc[2] += t2.x + t2.y;
      acc[3] += t3.x + t3.y; acc[4] += t4.x + t4.y; acc[5] += t5.x + t5.y;
    }
    // ---- y neighbours of tau through LDS (image st & 1)
    const int img = (st + 1) & 1;
    Tb[img][0][r][li] = t1;
    Tb[img][1][r][li] = t5;
    Tb[img][2][r][li] = t3;
    if (FULLROW) {
      if (l == 63) edge[img][0][r][zs] = t2.y;
      if (l == 0) {
        edge[img][1][r][zs] = t3.x;
        edge[img][2][r][zs] = t4.x;
      }
    }
    __syncthreads();
    const double2 t1yb = Tb[img][0][rm][li], t5yf = Tb[img][1][rp][li], t3yf = Tb[img][2][rp][li];
    double t2zb = prev_y(t2.y), t3zf = next_x(t3.x), t4zf = next_x(t4.x);
    if (FULLROW) {
      if (l == 0) t2zb = edge[img][0][r][zprev];
      if (l == 63) {
        t3zf = edge[img][1][r][znext];
        t4zf = edge[img][2][r][znext];
      }
    }
    // ---- divergence: f0 of this plane, f1 / f2 of the previous one
    if (own) {
      if (inside) {
        double2 f0;
        f0.x = (t0.x - t0m.x) * hx + (t5yf.x - t5.x) * hy + (t4.y - t4.x) * hz;
        f0.y = (t0.y - t0m.y) * hx + (t5yf.y - t5.y) * hy + (t4zf - t4.y) * hz;
        store_f(fo.p[0], oq, f0);
      }
      if (st >= 1) {
        const long op = plane(q - 1);
        store_f(fo.p[1], op, make_double2((t5.x - t5m.x) * hx + part1.x, (t5.y - t5m.y) * hx + part1.y));
        store_f(fo.p[2], op, make_double2((t4.x - t4m.x) * hx + part2.x, (t4.y - t4m.y) * hx + part2.y));
      }
    }
    part1.x = (t1.x - t1yb.x) * hy + (t3.y - t3.x) * hz;
    part1.y = (t1.y - t1yb.y) * hy + (t3zf - t3.y) * hz;
    part2.x = (t3yf.x - t3.x) * hy + (t2.x - t2zb) * hz;
    part2.y = (t3yf.y - t3.y) * hy + (t2.y - t2.x) * hz;
    t0m = t0; t5m = t5; t4m = t4;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double a = acc[c];
    a += dpp_move<0x128>(a);
    a += dpp_move<0x124>(a);
    a += dpp_move<0x122>(a);
    a += dpp_move<0x121>(a);
    acc[c] = (read_lane(a, 0) + read_lane(a, 16)) + (read_lane(a, 32) + read_lane(a, 48));
  }
  if (l == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) red[wv * 6 + c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double a = 0.0;
    for (int w = 0; w < TYR * NZS; ++w) a += red[w * 6 + threadIdx.x];
    partial[(long)blockIdx.x * 6 + threadIdx.x] = a;
  }
}

// Scalar modes (heat / porous), fast variant of k_sc_sweep (fg_kernels_scalar.hip): the per-voxel effective
// conductivity a = sum_p phi_p mu_p is precomputed (k_effective_moduli stores it in the first moduli array with
// 2 mu_p := mu_p), z neighbours come from the adjacent lanes.  T_k -> sums of squares of g_k = E + grad+ T_k and
// f = div-((a - 2 mu0) g_k).
__global__ __launch_bounds__(kBlock) void k_sc_sweep_fast(Grid g, double beta, const double* T, const double* a, double* f,
                                                          Vec6 E, double* partial, Sweep sw) {
  __shared__ double smem[4 * 6];
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const long npairs = (long)g.nx * g.ny * g.nzc;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const BlockRun run = block_run((npairs + kBlock - 1) / kBlock);
  for (long it = 0; it < run.count; ++it) {
    const long pidx_raw = (run.first + it * run.stride) * kBlock + threadIdx.x;
    const long pidx = pidx_raw < npairs ? pidx_raw : npairs - 1;  // clamped lanes discard their result
    const PairPos p = pair_pos_tiled(pidx, g, sw);
    const bool valid = pidx_raw < npairs && p.k < g.nz;
    const bool second = p.k + 1 < g.nz;
    const int lane = threadIdx.x & 63;
    const bool prev_ok = lane > 0 && p.k > 0;
    const bool next_ok = lane < 63 && p.k + 2 < g.nz && pidx_raw + 1 < npairs;
    // x neighbours through Grid::xw_lo / xw_hi: periodic in a whole grid, the spare planes of T and a in an x-slab
    const long xf = (p.i + 1 == g.nx ? (long)(g.nx - g.xw_hi) - p.i : 1L) * g.nyzp;
    const long xb = (p.i == 0 ? (long)(g.xw_lo - 1) : -1L) * g.nyzp;
    const long yf = (p.j + 1 == g.ny ? -(long)(g.ny - 1) : 1L) * g.nzp;
    const long yb = (p.j == 0 ? (long)(g.ny - 1) : -1L) * g.nzp;
    const long ro = p.off - p.k;
    const int k = p.k;
    const int kb = k == 0 ? g.nz - 1 : k - 1;
    const int kf2 = (k + 2 >= g.nz) ? k + 2 - g.nz : k + 2;
    double eTm = 0, eam = 0, eTp = 0;
    if (!prev_ok) {
      eTm = T[ro + kb];
      eam = a[ro + kb];
    }
    if (!next_ok) eTp = T[ro + kf2];
    const long rk = ro + k;
    double2 Tc = ld2(T, rk), Txf = ld2(T, rk + xf), Txb = ld2(T, rk + xb), Tyf = ld2(T, rk + yf), Tyb = ld2(T, rk + yb);
    double2 ac = ld2(a, rk), axb = ld2(a, rk + xb), ayb = ld2(a, rk + yb);
    if (!second) {  // odd nz, last pair: z+1 wraps to 0
      Tc.y = T[ro];
      Txf.y = T[ro + xf]; Txb.y = T[ro + xb]; Tyf.y = T[ro + yf]; Tyb.y = T[ro + yb];
    }
    const double t_prev = dpp_move<0x138>(Tc.y), a_prev = dpp_move<0x138>(ac.y), t_next = dpp_move<0x130>(Tc.x);
    const double Tm = prev_ok ? t_prev : eTm;   // T at z = k-1
    const double am = prev_ok ? a_prev : eam;   // a at z = k-1
    const double Tp = next_ok ? t_next : eTp;   // T at z = k+2
    // voxel z = k
    const double g0 = E.v[0] + (Txf.x - Tc.x) * hx, g1 = E.v[1] + (Tyf.x - Tc.x) * hy, g2 = E.v[2] + (Tc.y - Tc.x) * hz;
    const double g0b = E.v[0] + (Tc.x - Txb.x) * hx, g1b = E.v[1] + (Tc.x - Tyb.x) * hy, g2b = E.v[2] + (Tc.x - Tm) * hz;
    const double c0 = ac.x + beta;
    const double f0 = (c0 * g0 - (axb.x + beta) * g0b) * hx + (c0 * g1 - (ayb.x + beta) * g1b) * hy +
                      (c0 * g2 - (am + beta) * g2b) * hz;
    // voxel z = k+1
    const double h0 = E.v[0] + (Txf.y - Tc.y) * hx, h1 = E.v[1] + (Tyf.y - Tc.y) * hy, h2 = E.v[2] + (Tp - Tc.y) * hz;
    const double h0b = E.v[0] + (Tc.y - Txb.y) * hx, h1b = E.v[1] + (Tc.y - Tyb.y) * hy;
    const double c1 = ac.y + beta;
    const double f1 = (c1 * h0 - (axb.y + beta) * h0b) * hx + (c1 * h1 - (ayb.y + beta) * h1b) * hy +
                      (c1 * h2 - c0 * g2) * hz;   // the z-backward neighbour of k+1 is voxel k itself
    if (valid) {
      acc[0] += g0 * g0 + (second ? h0 * h0 : 0.0);
      acc[1] += g1 * g1 + (second ? h1 * h1 : 0.0);
      acc[2] += g2 * g2 + (second ? h2 * h2 : 0.0);
      st2(f, p.off, make_double2(f0, second ? f1 : 0.0));
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

// rows of at least 40 pairs: a row shorter than the 62 pairs of a tile is ONE tile whose surplus lanes hold wrapped-around
// copies (kp0 < 0) -- old / new library in one job: sweep at 80^3 20.9 -> 17.5 us, 96^3 29.1 -> 19.5, 100^3 32.0 -> 22.0,
// 112^3 41.3 -> 24.2, 120^3 47.0 -> 27.0 (100^3 8 630 -> 9 360 it/s, 120^3 5 990 -> 6 750); at 64^3 (32 pairs) the chunked
// sweep stays ahead (15.4 against 16.4 us)
bool u_tile_supported(const Grid& g) {
  const int nzh = g.nz / 2;
  return g.nz % 2 == 0 && nzh >= 40 && g.ny >= 14 && g.nx >= 4;
}

// Planes per march of the tiled sweep.  A march of LX planes costs LX + 3 steps (pipeline fill), the workgroups are dealt
// to the CUs in rounds, and a CU with two workgroups runs each at half speed, so the sweep takes about
// ceil(tiles * ceil(nx / LX) / CUs) * (LX + 3) steps: the LX with the smallest product wins, the longer march on a tie.
// Measured against the fixed 32 / 16 of round 1: 128^3 16 -> 12 planes (242 workgroups on 256 CUs) 0.0366 -> 0.0347 ms,
// 11 planes (264: a second round) 0.049 ms; 256^3 32 -> 64 (exactly 256 workgroups) 0.210 -> 0.203 ms, 48 (384) 0.265 ms;
// 512^3 32 -> 64: -1 %; thin x-slabs 32 x 256 x 256 -> 8, 64 x 512 x 512 -> 16 as measured before (44.7 -> 39.9 us,
// 281 -> 240 us).
inline int march_length(int nx, long tiles, int cus) {
  int best = nx < 4 ? nx : 4;
  long best_cost = -1;
  for (int lx = 4; lx <= nx && lx <= 64; ++lx) {   // (128-plane marches: 384^3 0.727 -> 0.766 ms, 512^3 no gain: neighbouring tiles drift apart)
    const long groups = tiles * ((nx + lx - 1) / lx);
    const long cost = ((groups + cus - 1) / cus) * (lx + 3);
    if (best_cost < 0 || cost <= best_cost) best = lx, best_cost = cost;
  }
  return best;
}

template <int TYR, int ZS, bool SUMT, bool PHI2 = false, bool CGP = false>
void launch_u_tile_t(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& u, const FieldPtrs<2>& mod,
                     const FieldPtrs<3>& f, const Vec6& E, double* partial, double* sumsq6, hipStream_t s,
                     const PhaseLin& lin = PhaseLin{0, 0, 0, 0}, const CgDirection& cg = CgDirection{}) {
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2, TZU = ZS ? 64 * ZS : 62;
  const int nzh = g.nz / 2;
  const int nty = (g.ny + TYU - 1) / TYU, ntz = (nzh + TZU - 1) / TZU;
  const int cus = device_cu_count();
  int LX = march_length(g.nx, (long)nty * ntz, cus);   // (a sweep of fixed lengths 3 ... 10 at 100^3 - 150^3: none beats it)
  if (LX > g.nx) LX = g.nx;
  const int ntx = (g.nx + LX - 1) / LX;
  int nb = nty * ntz * ntx;
  if (nb >= 8) nb = ((nb + 7) / 8) * 8;
  const size_t lds = 6 * TYR * NZS * 64 * sizeof(double2);
  static PerDeviceOnce configured;
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_u_tile<TYR, ZS, SUMT, PHI2, CGP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  const int nt = 3.0 * (double)g.n * sizeof(double) > 256.0 * 1024 * 1024 ? 1 : 0;
  hipLaunchKernelGGL((k_u_tile<TYR, ZS, SUMT, PHI2, CGP>), dim3(nb), dim3(TYR * NZS * 64), lds, s, g, -2 * mu_0, -lambda_0, u, mod, f,
                     E, partial, nty, ntz, LX, nt, lin, cg);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, SUMT ? 12 : 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_u_tile(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& u, const FieldPtrs<2>& mod,
                   const FieldPtrs<3>& f, const Vec6& E, double* partial, double* sumsq6, hipStream_t s,
                   bool sum_tau, const PhaseTable* two_phase) {
  const int nzh = g.nz / 2;
  if (two_phase) {   // mod.p[0] is phi_1 of two complementary phases: the default tile shapes
    const PhaseLin lin = {2 * two_phase->mu[0] - 2 * mu_0, 2 * (two_phase->mu[1] - two_phase->mu[0]),
                          two_phase->lambda[0] - lambda_0, two_phase->lambda[1] - two_phase->lambda[0]};
#define FG_PHI(R, Z)                                                                                                  \
  do {                                                                                                                \
    if (sum_tau) launch_u_tile_t<R, Z, true, true>(g, mu_0, lambda_0, u, mod, f, E, partial, sumsq6, s, lin);        \
    else launch_u_tile_t<R, Z, false, true>(g, mu_0, lambda_0, u, mod, f, E, partial, sumsq6, s, lin);               \
  } while (0)
    if (nzh == 64) FG_PHI(8, 1);
    else if (nzh == 128) FG_PHI(6, 2);
    else FG_PHI(8, 0);
#undef FG_PHI
    return;
  }
  if (sum_tau) {   // sumsq6[0..5] sums of squares of the strain, sumsq6[6..11] sums of the polarisation
    if (nzh == 64) launch_u_tile_t<8, 1, true>(g, mu_0, lambda_0, u, mod, f, E, partial, sumsq6, s);
    else if (nzh == 128) launch_u_tile_t<6, 2, true>(g, mu_0, lambda_0, u, mod, f, E, partial, sumsq6, s);
    else launch_u_tile_t<8, 0, true>(g, mu_0, lambda_0, u, mod, f, E, partial, sumsq6, s);
    return;
  }
#define FG_TILE(R, Z) launch_u_tile_t<R, Z, false>(g, mu_0, lambda_0, u, mod, f, E, partial, sumsq6, s)
  if (nzh == 64) {          // a z row is one wave
    FG_TILE(8, 1);
  } else if (nzh == 128) {  // a z row is two waves: 6 rows x 2 segments (256^3: 0.268 ms against 0.325 ms with halo lanes;
    FG_TILE(6, 2);          // 8 x 2 = 16 waves in lock-step 0.38 ms, 4 x 2 0.31 ms)
  } else {                  // general: tiles of 62 pairs with halo lanes
    FG_TILE(8, 0);       // (12- and 16-row tiles move fewer halo bytes -- 512^3: 8.95 -> 8.5 GB -- in the same time: rounds 1 and 5)
  }
#undef FG_TILE
}

// k_u_tile with the search direction of the conjugate gradients formed on the fly: u := r + b u -> po (see CgDirection);
// the default tile shapes, no sums of tau
void launch_u_tile_cg(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& p_old, const FieldPtrs<3>& r,
                      const FieldPtrs<3>& p_new, const FieldPtrs<2>& mod, const FieldPtrs<3>& f, const Vec6& E, const double* sc,
                      int i_num, int i_den, double nvox, double small, double* partial, double* sumsq6, hipStream_t s,
                      const PhaseTable* two_phase) {
  CgDirection cg;
  for (int c = 0; c < 3; ++c) cg.r[c] = r.p[c], cg.po[c] = p_new.p[c];
  cg.sc = sc;
  cg.i_num = i_num;
  cg.i_den = i_den;
  cg.nvox = nvox;
  cg.small = small;
  const int nzh = g.nz / 2;
  PhaseLin lin = {0, 0, 0, 0};
  if (two_phase)
    lin = PhaseLin{2 * two_phase->mu[0] - 2 * mu_0, 2 * (two_phase->mu[1] - two_phase->mu[0]), two_phase->lambda[0] - lambda_0,
                   two_phase->lambda[1] - two_phase->lambda[0]};
#define FG_CGK(R, Z)                                                                                                            \
  do {                                                                                                                          \
    if (two_phase) launch_u_tile_t<R, Z, false, true, true>(g, mu_0, lambda_0, p_old, mod, f, E, partial, sumsq6, s, lin, cg);   \
    else launch_u_tile_t<R, Z, false, false, true>(g, mu_0, lambda_0, p_old, mod, f, E, partial, sumsq6, s, lin, cg);            \
  } while (0)
  if (nzh == 64) FG_CGK(8, 1);
  else if (nzh == 128) FG_CGK(6, 2);
  else FG_CGK(8, 0);
#undef FG_CGK
}

// *flag := 0 unless phi0 == 1 - phi1 bit for bit at every voxel (what normalizePhi F:17613-17626 produces for two phases);
// the caller sets *flag = 1 beforehand
__global__ __launch_bounds__(kBlock) void k_complement_check(Grid g, const double* phi0, const double* phi1, int* flag) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  bool bad = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (long)gridDim.x * blockDim.x) {
    const long row = i / g.nzc;
    const int k = 2 * (int)(i - row * g.nzc);
    if (k >= g.nz) continue;
    const double2 a = ld2(phi0, 2 * i), b = ld2(phi1, 2 * i);
    bad = bad || (a.x != 1.0 - b.x) || (k + 1 < g.nz && a.y != 1.0 - b.y);
  }
  if (bad) *flag = 0;
}

void launch_complement_check(const Grid& g, const double* phi0, const double* phi1, int* flag, hipStream_t s) {
  const long npairs = (long)g.nx * g.ny * g.nzc;
  long nb = (npairs + kBlock - 1) / kBlock;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(k_complement_check, dim3((unsigned)nb), dim3(kBlock), 0, s, g, phi0, phi1, flag);
  FG_HIP_CHECK(hipGetLastError());
}

// Tiled marching form of the scalar sweep, the one-component sibling of k_u_tile: a workgroup of TYR rows (one or ZS
// waves each, lanes = z pairs) marches along x.  x neighbours are the thread's own values of the previous / next plane,
// y neighbours (T of rows j-1 and j+1, the conductivity of row j-1) come from one LDS exchange per plane -- two
// images, so one barrier per step --, z neighbours from the adjacent lanes.  Rows 0 and TYR-1 and, for ZS = 0, lanes
// 0 and 63 are halo.  T_k -> sums of squares of g_k = E + grad+ T_k and f = div-((a - 2 mu0) g_k), every T and a value
// loaded once per tile (k_sc_sweep_fast: 8 loads per pair out of L2).
// SUMT: the three sums of the flux polarisation tau = (a - 2 mu0) g as well (partial slots 3..5): <tau> drives the mixed
// boundary conditions (initBCProjector F:20228-20239 in GammaOperatorStaggeredHeat F:20342-20350)
// CGP: T := r + b T, the new search direction of the conjugate gradients in potential space, formed on the fly and stored to
// po where this thread owns the voxels (see CgDirection at k_u_tile)
struct ScCgDirection {
  const double* r;
  double* po;
  const double* sc;
  int i_num, i_den;
  double nvox, small;
};

template <int TYR, int ZS, bool SUMT = false, bool CGP = false>
__global__ __launch_bounds__(TYR * (ZS ? ZS : 1) * 64) void k_sc_tile(Grid g, double beta, const double* T, const double* a,
                                                                       double* fo, Vec6 E, double* partial, int nty, int ntz,
                                                                       int LX, int nt, ScCgDirection cg) {
  constexpr bool FULLROW = ZS > 0;
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2;
  constexpr int TZU = FULLROW ? 64 * NZS : 62;
  constexpr int RW = NZS * 64;
  __shared__ double2 Tb[2][TYR][RW];
  __shared__ double2 Cb[2][TYR][RW];
  constexpr int NA = SUMT ? 6 : 3;
  __shared__ double red[TYR * NZS * NA];

  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r = wv / NZS, zs = wv % NZS;
  const int li = zs * 64 + l;
  const int zprev = (zs + NZS - 1) % NZS, znext = (zs + 1) % NZS;
  const int nzh = g.nz / 2;
  int b = blockIdx.x;
  {
    const int nb = gridDim.x;
    if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;   // one XCD takes a contiguous run of tiles
  }
  const int tz = b % ntz;
  b /= ntz;
  const int ty = b % nty;
  const int tx = b / nty;
  const bool surplus = tx * LX >= g.nx;
  const int j0 = min(ty * TYU, g.ny - TYU), kp0 = min(tz * TZU, nzh - TZU), x0 = surplus ? 0 : tx * LX;
  const int jr = j0 - 1 + r;
  const int j = jr < 0 ? jr + g.ny : (jr >= g.ny ? jr - g.ny : jr);
  const int kr = FULLROW ? li : kp0 - 1 + l;
  const int kp = kr < 0 ? kr + nzh : (kr >= nzh ? kr - nzh : kr);
  const bool own = r >= 1 && r <= TYU && jr >= ty * TYU && (FULLROW || (l >= 1 && l <= TZU && kr >= tz * TZU));
  const long rowoff = (long)j * g.nzp + 2 * kp;
  const int rm = r > 0 ? r - 1 : 0, rp = r + 1 < TYR ? r + 1 : TYR - 1;
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const int nsteps = surplus ? 0 : (x0 + LX <= g.nx ? LX : g.nx - x0);

  auto plane = [&](int q) {   // x neighbours through Grid::xw_lo / xw_hi (periodic grid: both nx; x-slab: its spare planes)
    const int x = q < 0 ? q + g.xw_lo : (q >= g.nx ? q - g.xw_hi : q);
    return (long)x * g.nyzp + rowoff;
  };
  auto store_f = [&](long off, double2 v) {
    if (nt) {
      typedef double fg_v2d __attribute__((ext_vector_type(2)));
      fg_v2d t;
      t.x = v.x;
      t.y = v.y;
      __builtin_nontemporal_store(t, reinterpret_cast<fg_v2d*>(fo + off));
    } else {
      st2(fo, off, v);
    }
  };

  const double cgb = CGP ? (cg.sc[cg.i_num] / cg.nvox + cg.small) / (cg.sc[cg.i_den] / cg.nvox + cg.small) : 0.0;
  auto load_T = [&](long o, bool keep) {
    double2 v = ld2(T, o);
    if (CGP) {
      const double2 rv = ld2(cg.r, o);
      v.x = rv.x + cgb * v.x;
      v.y = rv.y + cgb * v.y;
      if (keep && own) st2(cg.po, o, v);
    }
    return v;
  };
  // plane x0 - 1 gives the x flux entering plane x0
  double2 Tc = load_T(plane(x0 - 1), false), Tn = load_T(plane(x0), nsteps > 0), T2 = load_T(plane(x0 + 1), nsteps > 1);
  double2 ac = ld2(a, plane(x0 - 1)), an = ld2(a, plane(x0));
  double2 q0m;   // x flux of the previous plane
  q0m.x = (ac.x + beta) * (E.v[0] + (Tn.x - Tc.x) * hx);
  q0m.y = (ac.y + beta) * (E.v[0] + (Tn.y - Tc.y) * hx);
  Tc = Tn; Tn = T2; ac = an;
  double acc[NA] = {};

  for (int st = 0; st < nsteps; ++st) {
    const int q = x0 + st;
    const long oq = plane(q);
    T2 = load_T(plane(q + 2), st + 2 < nsteps);   // two planes ahead of T, one of a (the last step's are unused but in range)
    an = ld2(a, plane(q + 1));
    const double2 cc = make_double2(ac.x + beta, ac.y + beta);
    const int img = st & 1;
    Tb[img][r][li] = Tc;
    Cb[img][r][li] = cc;
    __syncthreads();
    const double2 Tyf = Tb[img][rp][li], Tyb = Tb[img][rm][li], cyb = Cb[img][rm][li];
    double Tzb = dpp_move<0x138>(Tc.y), czb = dpp_move<0x138>(cc.y), Tzf = dpp_move<0x130>(Tc.x);
    if (FULLROW) {
      if (l == 0) {
        Tzb = Tb[img][r][zprev * 64 + 63].y;
        czb = Cb[img][r][zprev * 64 + 63].y;
      }
      if (l == 63) Tzf = Tb[img][r][znext * 64].x;
    }
    // gradient at the two voxels (forward differences) and the backward-neighbour fluxes
    const double g0x = E.v[0] + (Tn.x - Tc.x) * hx, g0y = E.v[0] + (Tn.y - Tc.y) * hx;
    const double g1x = E.v[1] + (Tyf.x - Tc.x) * hy, g1y = E.v[1] + (Tyf.y - Tc.y) * hy;
    const double g2x = E.v[2] + (Tc.y - Tc.x) * hz, g2y = E.v[2] + (Tzf - Tc.y) * hz;
    const double q0x = cc.x * g0x, q0y = cc.y * g0y;
    const double q1bx = cyb.x * (E.v[1] + (Tc.x - Tyb.x) * hy), q1by = cyb.y * (E.v[1] + (Tc.y - Tyb.y) * hy);
    const double q2bx = czb * (E.v[2] + (Tc.x - Tzb) * hz);
    double2 f;
    f.x = (q0x - q0m.x) * hx + (cc.x * g1x - q1bx) * hy + (cc.x * g2x - q2bx) * hz;
    f.y = (q0y - q0m.y) * hx + (cc.y * g1y - q1by) * hy + (cc.y * g2y - cc.x * g2x) * hz;
    if (own) {
      acc[0] += g0x * g0x + g0y * g0y;
      acc[1] += g1x * g1x + g1y * g1y;
      acc[2] += g2x * g2x + g2y * g2y;
      if (SUMT) {
        acc[NA - 3] += q0x + q0y;
        acc[NA - 2] += cc.x * g1x + cc.y * g1y;
        acc[NA - 1] += cc.x * g2x + cc.y * g2y;
      }
      store_f(oq, f);
    }
    q0m.x = q0x; q0m.y = q0y;
    Tc = Tn; Tn = T2; ac = an;
  }
#pragma unroll
  for (int c = 0; c < NA; ++c) {
    double v = acc[c];
    v += dpp_move<0x128>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x122>(v);
    v += dpp_move<0x121>(v);
    acc[c] = (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
  }
  if (l == 0) {
#pragma unroll
    for (int c = 0; c < NA; ++c) red[wv * NA + c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = 0.0;
    if (threadIdx.x < NA)
      for (int w = 0; w < TYR * NZS; ++w) v += red[w * NA + threadIdx.x];
    partial[(long)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// sums of the scalar sweep with SUMT: [0..2] norms, [3..5] tau -> sumsq6 = (norms, 0, 0, 0), sumtau3 = tau sums
__global__ void k_sc_split_sums(double* sumsq6, double* sumtau3) {
  const int c = threadIdx.x;
  if (c < 3) {
    sumtau3[c] = sumsq6[3 + c];
    sumsq6[3 + c] = 0.0;
  }
}

template <int TYR, int ZS>
void launch_sc_tile_t(const Grid& g, double mu_0, const double* T, const double* a, double* f, const Vec6& E, double* partial,
                      double* sumsq6, hipStream_t s, double* sumtau3, const ScCgDirection* cgd = nullptr) {
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2, TZU = ZS ? 64 * ZS : 62;
  const int nzh = g.nz / 2;
  const int nty = (g.ny + TYU - 1) / TYU, ntz = (nzh + TZU - 1) / TZU;
  const int cus = device_cu_count();
  int LX = march_length(g.nx, (long)nty * ntz, 2 * cus);   // the light scalar sweep runs two workgroups per CU at full speed
  if (LX > g.nx) LX = g.nx;
  const int ntx = (g.nx + LX - 1) / LX;
  int nb = nty * ntz * ntx;
  if (nb >= 8) nb = ((nb + 7) / 8) * 8;
  const int nt = (double)g.n * sizeof(double) > 128.0 * 1024 * 1024 ? 1 : 0;
  if (sumtau3) {
    hipLaunchKernelGGL((k_sc_tile<TYR, ZS, true>), dim3(nb), dim3(TYR * NZS * 64), 0, s, g, -2 * mu_0, T, a, f, E, partial, nty,
                       ntz, LX, nt, ScCgDirection{});
    FG_HIP_CHECK(hipGetLastError());
    fold_sum(partial, nb, 6, sumsq6, s);
    hipLaunchKernelGGL(k_sc_split_sums, dim3(1), dim3(64), 0, s, sumsq6, sumtau3);
    FG_HIP_CHECK(hipGetLastError());
    return;
  }
  if (cgd)
    hipLaunchKernelGGL((k_sc_tile<TYR, ZS, false, true>), dim3(nb), dim3(TYR * NZS * 64), 0, s, g, -2 * mu_0, T, a, f, E, partial, nty,
                       ntz, LX, nt, *cgd);
  else
    hipLaunchKernelGGL((k_sc_tile<TYR, ZS>), dim3(nb), dim3(TYR * NZS * 64), 0, s, g, -2 * mu_0, T, a, f, E, partial, nty, ntz,
                       LX, nt, ScCgDirection{});
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

// Scalar sibling of k_cgu_tile (conjugate gradients in potential space, k_sc_cg_dot / k_sc_cg_axpy of fg_kernels_scalar.hip):
// gradients are forward differences, so a tile needs the NEXT plane (registers), the next row (LDS) and the next lane (DPP).
//   MODE 0:  partial[0] = sum grad A . (grad A - grad B),  A = a, B = b
//   MODE 1:  A = a + alpha y -> ao,  B = b - alpha (y - w) -> bo (out of place),  partial[0..2] = sums of (E + grad A)_c^2,
//            partial[6] = sum grad B . grad B
template <int TYR, int ZS, int MODE>
__global__ __launch_bounds__(TYR * (ZS ? ZS : 1) * 64) void k_sc_cgu_tile(Grid g, const double* a, const double* b, const double* y,
                                                                           const double* w, double* ao, double* bo, Vec6 E,
                                                                           const double* sc, int i_num, int i_den, double nvox,
                                                                           double small, double* partial, int nty, int ntz, int LX) {
  constexpr bool FULLROW = ZS > 0;
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2;
  constexpr int TZU = FULLROW ? 64 * NZS : 62;
  constexpr int RW = NZS * 64;
  constexpr int NS = 7;
  __shared__ double2 Xb[2][2][TYR][RW];   // [image][A, B][row][pair]
  __shared__ double red[TYR * NZS * NS];
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r = wv / NZS, zs = wv % NZS;
  const int li = zs * 64 + l;
  const int znext = (zs + 1) % NZS;
  const int nzh = g.nz / 2;
  int bi = blockIdx.x;
  {
    const int nb = gridDim.x;
    if (nb % 8 == 0) bi = (bi % 8) * (nb / 8) + bi / 8;
  }
  const int tz = bi % ntz;
  bi /= ntz;
  const int ty = bi % nty;
  const int tx = bi / nty;
  const bool surplus = tx * LX >= g.nx;
  const int j0 = min(ty * TYU, g.ny - TYU), kp0 = min(tz * TZU, nzh - TZU), x0 = surplus ? 0 : tx * LX;
  const int jr = j0 - 1 + r;
  const int j = jr < 0 ? jr + g.ny : (jr >= g.ny ? jr - g.ny : jr);
  const int kr = FULLROW ? li : kp0 - 1 + l;
  const int kp = kr < 0 ? kr + nzh : (kr >= nzh ? kr - nzh : kr);
  const bool own = r >= 1 && r <= TYU && jr >= ty * TYU && (FULLROW || (l >= 1 && l <= TZU && kr >= tz * TZU));
  const long rowoff = (long)j * g.nzp + 2 * kp;
  const int rp = r + 1 < TYR ? r + 1 : TYR - 1;
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  const int nsteps = surplus ? 0 : (x0 + LX <= g.nx ? LX : g.nx - x0);
  const double al = MODE == 1 ? (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small) : 0.0;
  auto plane = [&](int q) {
    const int x = q < 0 ? q + g.xw_lo : (q >= g.nx ? q - g.xw_hi : q);
    return (long)x * g.nyzp + rowoff;
  };
  auto fetch = [&](int q, double2& A, double2& B, bool keep) {
    const long o = plane(q);
    A = ld2(a, o);
    B = ld2(b, o);
    if (MODE == 1) {
      const double2 yv = ld2(y, o), wv2 = ld2(w, o);
      A.x = A.x + al * yv.x;
      A.y = A.y + al * yv.y;
      B.x = B.x - al * (yv.x - wv2.x);
      B.y = B.y - al * (yv.y - wv2.y);
      if (keep && own) {
        st2(ao, o, A);
        st2(bo, o, B);
      }
    }
  };
  double acc[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) acc[c] = 0.0;
  if (nsteps > 0) {
    double2 Ac, Bc, An, Bn;
    fetch(x0, Ac, Bc, true);
    fetch(x0 + 1, An, Bn, nsteps > 1);
    for (int st = 0; st < nsteps; ++st) {
      const int img = st & 1;
      Xb[img][0][r][li] = Ac;
      Xb[img][1][r][li] = Bc;
      __syncthreads();
      const double2 Ayf = Xb[img][0][rp][li], Byf = Xb[img][1][rp][li];
      double Azf = dpp_move<0x130>(Ac.x), Bzf = dpp_move<0x130>(Bc.x);
      if (FULLROW && l == 63) {
        Azf = Xb[img][0][r][znext * 64].x;
        Bzf = Xb[img][1][r][znext * 64].x;
      }
      if (own) {
        const double ga0x = (An.x - Ac.x) * hx, ga0y = (An.y - Ac.y) * hx;
        const double ga1x = (Ayf.x - Ac.x) * hy, ga1y = (Ayf.y - Ac.y) * hy;
        const double ga2x = (Ac.y - Ac.x) * hz, ga2y = (Azf - Ac.y) * hz;
        const double gb0x = (Bn.x - Bc.x) * hx, gb0y = (Bn.y - Bc.y) * hx;
        const double gb1x = (Byf.x - Bc.x) * hy, gb1y = (Byf.y - Bc.y) * hy;
        const double gb2x = (Bc.y - Bc.x) * hz, gb2y = (Bzf - Bc.y) * hz;
        if (MODE == 0) {
          acc[0] += (ga0x * (ga0x - gb0x) + ga1x * (ga1x - gb1x) + ga2x * (ga2x - gb2x)) +
                    (ga0y * (ga0y - gb0y) + ga1y * (ga1y - gb1y) + ga2y * (ga2y - gb2y));
        } else {
          const double e0x = E.v[0] + ga0x, e0y = E.v[0] + ga0y, e1x = E.v[1] + ga1x, e1y = E.v[1] + ga1y;
          const double e2x = E.v[2] + ga2x, e2y = E.v[2] + ga2y;
          acc[0] += e0x * e0x + e0y * e0y;
          acc[1] += e1x * e1x + e1y * e1y;
          acc[2] += e2x * e2x + e2y * e2y;
          acc[6] += (gb0x * gb0x + gb1x * gb1x + gb2x * gb2x) + (gb0y * gb0y + gb1y * gb1y + gb2y * gb2y);
        }
      }
      Ac = An;
      Bc = Bn;
      if (st + 1 < nsteps) fetch(x0 + st + 2, An, Bn, st + 2 < nsteps);
    }
  }
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    double v = acc[c];
    v += dpp_move<0x128>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x122>(v);
    v += dpp_move<0x121>(v);
    acc[c] = (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
  }
  if (l == 0) {
#pragma unroll
    for (int c = 0; c < NS; ++c) red[wv * NS + c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double v = 0.0;
    for (int q = 0; q < TYR * NZS; ++q) v += red[q * NS + threadIdx.x];
    partial[(long)blockIdx.x * NS + threadIdx.x] = v;
  }
}

template <int TYR, int ZS>
void launch_sc_cgu_tile_t(int mode, const Grid& g, const double* a, const double* b, const double* y, const double* w, double* ao,
                          double* bo, const Vec6& E, const double* sc, int i_num, int i_den, double nvox, double small,
                          double* partial, double* out7, hipStream_t s) {
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2, TZU = ZS ? 64 * ZS : 62;
  const int nzh = g.nz / 2;
  const int nty = (g.ny + TYU - 1) / TYU, ntz = (nzh + TZU - 1) / TZU;
  int LX = march_length(g.nx, (long)nty * ntz, 2 * device_cu_count());
  if (LX > g.nx) LX = g.nx;
  const int ntx = (g.nx + LX - 1) / LX;
  int nb = nty * ntz * ntx;
  if (nb >= 8) nb = ((nb + 7) / 8) * 8;
  if (mode == 0)
    hipLaunchKernelGGL((k_sc_cgu_tile<TYR, ZS, 0>), dim3(nb), dim3(TYR * NZS * 64), 0, s, g, a, b, y, w, ao, bo, E, sc, i_num, i_den, nvox,
                       small, partial, nty, ntz, LX);
  else
    hipLaunchKernelGGL((k_sc_cgu_tile<TYR, ZS, 1>), dim3(nb), dim3(TYR * NZS * 64), 0, s, g, a, b, y, w, ao, bo, E, sc, i_num, i_den, nvox,
                       small, partial, nty, ntz, LX);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 7, out7, s);
  FG_HIP_CHECK(hipGetLastError());
}

// the scalar modes' CG vector sweeps in the tiled form (grids of sc_sweep_tiled), see k_sc_cgu_tile
void launch_sc_cgu_tile(int mode, const Grid& g, const double* a, const double* b, const double* y, const double* w, double* ao,
                        double* bo, const Vec6& E, const double* sc, int i_num, int i_den, double nvox, double small, double* partial,
                        double* out7, hipStream_t s) {
  const int nzh = g.nz / 2;
  if (nzh == 64) launch_sc_cgu_tile_t<8, 1>(mode, g, a, b, y, w, ao, bo, E, sc, i_num, i_den, nvox, small, partial, out7, s);
  else if (nzh == 128) launch_sc_cgu_tile_t<6, 2>(mode, g, a, b, y, w, ao, bo, E, sc, i_num, i_den, nvox, small, partial, out7, s);
  else launch_sc_cgu_tile_t<8, 0>(mode, g, a, b, y, w, ao, bo, E, sc, i_num, i_den, nvox, small, partial, out7, s);
}

// the tiled scalar sweep on the new search direction T_p = T_r + b T_p (b from the device sums), stored to p_new
void launch_sc_sweep_cg(const Grid& g, double mu_0, const double* p_old, const double* r, double* p_new, const double* a, double* f,
                        const Vec6& E, const double* sc, int i_num, int i_den, double nvox, double small, double* partial,
                        double* sumsq6, hipStream_t s) {
  ScCgDirection cg = {r, p_new, sc, i_num, i_den, nvox, small};
  const int nzh = g.nz / 2;
  if (nzh == 64) launch_sc_tile_t<8, 1>(g, mu_0, p_old, a, f, E, partial, sumsq6, s, nullptr, &cg);
  else if (nzh == 128) launch_sc_tile_t<6, 2>(g, mu_0, p_old, a, f, E, partial, sumsq6, s, nullptr, &cg);
  else launch_sc_tile_t<8, 0>(g, mu_0, p_old, a, f, E, partial, sumsq6, s, nullptr, &cg);
}

bool sc_sweep_tiled(const Grid& g) {
  return u_tile_supported(g);
}

bool launch_sc_sweep_fast(const Grid& g, double mu_0, const double* T, const double* a, double* f, const Vec6& E,
                          double* partial, double* sumsq6, hipStream_t s, double* sumtau3) {
  if (sc_sweep_tiled(g)) {
    const int nzh = g.nz / 2;
    if (nzh == 64) launch_sc_tile_t<8, 1>(g, mu_0, T, a, f, E, partial, sumsq6, s, sumtau3);
    else if (nzh == 128) launch_sc_tile_t<6, 2>(g, mu_0, T, a, f, E, partial, sumsq6, s, sumtau3);   // 48 KB of LDS images
    else launch_sc_tile_t<8, 0>(g, mu_0, T, a, f, E, partial, sumsq6, s, sumtau3);
    return sumtau3 != nullptr;
  }
  const int nb = sweep_blocks((long)g.nx * g.ny * g.nzc);
  hipLaunchKernelGGL(k_sc_sweep_fast, dim3(nb), dim3(kBlock), 0, s, g, -2 * mu_0, T, a, f, E, partial, chunk_rows(g));
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
  return false;   // the untiled sweep carries no sums of tau
}

template <int TYR, int ZS>
void launch_eps_tile_t(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<6>& eps, const FieldPtrs<2>& mod,
                       const FieldPtrs<3>& f, double* partial, double* sum6, hipStream_t s) {
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2, TZU = ZS ? 64 * ZS : 62;
  const int nzh = g.nz / 2;
  const int nty = (g.ny + TYU - 1) / TYU, ntz = (nzh + TZU - 1) / TZU;
  const int cus = device_cu_count();
  int LX = 32;
  if ((long)nty * ntz * ((g.nx + 31) / 32) < cus) LX = 16;
  if (LX > g.nx) LX = g.nx;
  const int ntx = (g.nx + LX - 1) / LX;
  int nb = nty * ntz * ntx;
  if (nb >= 8) nb = ((nb + 7) / 8) * 8;
  const size_t lds = 6 * TYR * NZS * 64 * sizeof(double2);
  static PerDeviceOnce configured;
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eps_tile<TYR, ZS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  const int nt = 3.0 * (double)g.n * sizeof(double) > 256.0 * 1024 * 1024 ? 1 : 0;
  hipLaunchKernelGGL((k_eps_tile<TYR, ZS>), dim3(nb), dim3(TYR * NZS * 64), lds, s, g, -2 * mu_0, -lambda_0, eps, mod, f,
                     partial, nty, ntz, LX, nt);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 6, sum6, s);
  FG_HIP_CHECK(hipGetLastError());
}

void launch_eps_tile(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<6>& eps, const FieldPtrs<2>& mod,
                     const FieldPtrs<3>& f, double* partial, double* sum6, hipStream_t s) {
  const int nzh = g.nz / 2;
  if (nzh == 64) launch_eps_tile_t<8, 1>(g, mu_0, lambda_0, eps, mod, f, partial, sum6, s);
  else if (nzh == 128) launch_eps_tile_t<6, 2>(g, mu_0, lambda_0, eps, mod, f, partial, sum6, s);
  else launch_eps_tile_t<8, 0>(g, mu_0, lambda_0, eps, mod, f, partial, sum6, s);
}

template <int TYR, int ZS, int MODE>
void launch_cgu_tile_t(const Grid& g, const FieldPtrs<3>& a, const FieldPtrs<3>& b, const FieldPtrs<3>& y, const FieldPtrs<3>& w,
                       const FieldPtrs<3>& ao, const FieldPtrs<3>& bo, const Vec6& E, const double* sc, int i_num, int i_den,
                       double nvox, double small, double* partial, double* out7, hipStream_t s) {
  constexpr int NZS = ZS ? ZS : 1;
  constexpr int TYU = TYR - 2, TZU = ZS ? 64 * ZS : 62;
  const int nzh = g.nz / 2;
  const int nty = (g.ny + TYU - 1) / TYU, ntz = (nzh + TZU - 1) / TZU;
  int LX = march_length(g.nx, (long)nty * ntz, device_cu_count());
  if (LX > g.nx) LX = g.nx;
  const int ntx = (g.nx + LX - 1) / LX;
  int nb = nty * ntz * ntx;
  if (nb >= 8) nb = ((nb + 7) / 8) * 8;
  const size_t lds = 6 * TYR * NZS * 64 * sizeof(double2);
  static PerDeviceOnce configured;
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cgu_tile<TYR, ZS, MODE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  const int nt = (MODE == 1 && 3.0 * (double)g.n * sizeof(double) > 256.0 * 1024 * 1024) ? 1 : 0;
  hipLaunchKernelGGL((k_cgu_tile<TYR, ZS, MODE>), dim3(nb), dim3(TYR * NZS * 64), lds, s, g, a, b, y, w, ao, bo, E, sc, i_num, i_den,
                     nvox, small, partial, nty, ntz, LX, nt);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, 7, out7, s);
  FG_HIP_CHECK(hipGetLastError());
}

// mode 0: out7[0] = a : (a - b) in the gradient inner product (y, w, ao, bo, sc unused); mode 1: the update
// A = a + alpha y, B = b - alpha (y - w) -> ao, bo (buffers other than a, b, y, w) with out7 = norms of E + grad_s A, B : B
void launch_cgu_tile(int mode, const Grid& g, const FieldPtrs<3>& a, const FieldPtrs<3>& b, const FieldPtrs<3>& y,
                     const FieldPtrs<3>& w, const FieldPtrs<3>& ao, const FieldPtrs<3>& bo, const Vec6& E, const double* sc,
                     int i_num, int i_den, double nvox, double small, double* partial, double* out7, hipStream_t s) {
  const int nzh = g.nz / 2;
#define FG_CGT(R, Z)                                                                                                          \
  do {                                                                                                                        \
    if (mode == 0) launch_cgu_tile_t<R, Z, 0>(g, a, b, y, w, ao, bo, E, sc, i_num, i_den, nvox, small, partial, out7, s);       \
    else launch_cgu_tile_t<R, Z, 1>(g, a, b, y, w, ao, bo, E, sc, i_num, i_den, nvox, small, partial, out7, s);                 \
  } while (0)
  if (nzh == 64) FG_CGT(8, 1);
  else if (nzh == 128) FG_CGT(6, 2);
  else FG_CGT(8, 0);
#undef FG_CGT
}

}  // namespace fg
