// The displacement sweep with BOTH z transforms attached (round 4): the loop state is the z half spectrum of u_k (what the
// inverse y pass leaves), and the sweep hands the z half spectrum of f_{k+1} to the forward y pass.
//
//   fftInvVector's last pass (c2r along z, F:18513-18528 / F:7239-7244)  -> in LDS
//   epsOperatorStaggered F:18614-18692 + component_norm F:10127 of pass k,
//   calcStress / Voigt PK1 F:18134, F:12752 + divOperatorStaggered F:18853 of pass k+1   (k_u_tile's arithmetic)
//   fftVector's first pass (r2c along z, F:18481-18500 / F:7232-7237)     <- from LDS
//
// so neither u nor f exists in real space in memory: per voxel the sweep reads 24 B of spectrum + 8 B of phase fraction and
// writes 24 B of spectrum (56 B, k_u_tile's own figure) where c2r + sweep + r2c move 48 + 56 + 48 = 152 B.
//
// A workgroup owns TY = NR - 2 whole z rows (rows 0 and NR - 1 of the tile are halo) and marches along x like k_u_tile.
// Thread roles (one thread per z pair of the tile, NR * M threads, M = nz / 2):
//   stencil (all threads)   row r = tid / M, pair li = tid % M: strain / polarisation of plane q (step B1), divergence (B2)
//   c2r     (tid < 3 NR T)  T = M / 4 threads per line, FOUR points each (Line4<M>: radix 4 / 2 -- the 8-point radix-8 lines
//                           of the FFT passes need ~100 VGPRs on top of the sweep's own state): plane q + 2 -> U image,
//                           spectrum of plane q + 3 requested right after the merge (one step ahead)
//   r2c     (the other waves) the finished f lines, a few at a time: -> half spectra -> memory
// The T threads of a line sit in one wave, so the exchanges between the radix passes need wave-local fences only, and a
// line exchanges IN PLACE in its own LDS region (a wave's LDS queue is in order).  Per step three workgroup barriers:
//   B1 | barrier | B2 | barrier | c2r of the plane after next BESIDE the r2c of the finished lines | barrier
// (the transforms are two thirds of a step's instructions: with the r2c behind the c2r's barrier, as first built, the waves
// holding both roles ran them back to back while the others idled -- 18 k cycles per step, tools/uz_probe.hip)
// LDS (UzGeom): F image (3 TY line regions in the FFT's padded re / im layout, written by B2) | two U images (3 NR line
//      regions each: c2r scratch, then the natural-order real rows; planes q and q + 1 -- the stencil reads plane q and its
//      y / z neighbours from LDS instead of carrying them in registers across the transforms) | tau exchange [3][NR][M] pairs |
//      pass twiddles | unit roots | edges | sums.
#include "fg_kernels.h"

#include <cstdlib>
#include <type_traits>

#include "fg_fft_core.h"
#include "fg_hip_util.h"
#include "fg_kernels_common.h"

namespace fg {

namespace {

using namespace fft;

// tools/uz_probe.hip compiles this file with -DFG_PROBE_UZ: cycle stamps of three waves (the first and a middle c2r wave, the
// last = r2c wave) of sampled workgroups inside one marching step.  Empty in the library.
#ifdef FG_PROBE_UZ
constexpr int kUzProbeBlocks = 64, kUzProbeSlots = 16;
__device__ unsigned long long g_uz_probe[kUzProbeBlocks][3][kUzProbeSlots];
#define FG_UZ_MARK(slot)                                                                                          \
  do {                                                                                                            \
    if (st == FG_PROBE_UZ_STEP && (tid & 63) == 0 && blockIdx.x % FG_PROBE_UZ_STRIDE == 0 &&                      \
        blockIdx.x / FG_PROBE_UZ_STRIDE < kUzProbeBlocks) {                                                       \
      const int w_ = tid >> 6, nw_ = (int)blockDim.x >> 6;                                                        \
      const int k_ = w_ == 0 ? 0 : (w_ == nw_ / 2 + 1 ? 1 : (w_ == nw_ - 1 ? 2 : -1));                            \
      if (k_ >= 0) g_uz_probe[blockIdx.x / FG_PROBE_UZ_STRIDE][k_][(slot)] = __builtin_readcyclecounter();        \
    }                                                                                                             \
  } while (0)
#else
#define FG_UZ_MARK(slot) do { } while (0)
#endif
#ifndef FG_UZ_ABLATE   // probe builds: 1 = no spectrum stores, 2 = no spectrum prefetch (timing experiments, wrong results)
#define FG_UZ_ABLATE 0
#endif

struct PhaseLin {   // see k_u_tile: moduli of two complementary phases from phi_1
  double a0, da, b0, db;
};

constexpr int kUzDumpGroups = 256;   // dump slots for stores that are not due: one per thread of this many workgroups

template <int M, int NR>
struct UzGeom {
  static constexpr int TY = NR - 2;
  static constexpr int NZS = M / 64;            // waves per z row
  static constexpr int NTH = NR * M;
  static constexpr int T = M / 4;               // FFT threads per line (four points each)
  static constexpr int NLA = 3 * NR, NLC = 3 * TY;
  static constexpr int LS = M + M / 8 + 2;      // doubles of the re (im) plane of a line
  static constexpr int ROWD = 2 * LS;           // doubles per line region
  static constexpr int A_THREADS = NLA * T, C_THREADS = NLC * T;
  static constexpr int C0 = A_THREADS;                        // the remaining waves take the r2c lines, LPR at a time
  static constexpr int LPR = (NTH - A_THREADS) / T, NRND = (NLC + LPR - 1) / LPR;
  // LDS order: F image | U images | tau exchange | twiddles | unit roots | edges | sums.  The stencil addresses the rows
  // r - 1, r, r + 1 of a buffer as ONE base (row r - 1) plus constant offsets; for the halo rows of the tile one of those
  // rows lies outside the buffer -- in the buffer before / behind it, whose contents then feed values nobody consumes.
  static constexpr int F_OFF = 0;
  // (one spare line region behind the F image: where the threads of the last, partly empty r2c round transform nothing)
  static constexpr int U_OFF = F_OFF + ROWD * (NLC + (NLC % LPR ? 1 : 0)), U_IMG = ROWD * NLA;   // two images: planes q and q + 1
  static constexpr int TB_OFF = U_OFF + 2 * U_IMG;
  static constexpr int TW_OFF = TB_OFF + 3 * NR * M * 2;
  static constexpr int WZ_OFF = TW_OFF + 2 * tw_total4(M);
  static constexpr int EDGE_OFF = WZ_OFF + 2 * T;
  static constexpr int RED_OFF = EDGE_OFF + 3 * NR * NZS;
  static constexpr int LDS_DOUBLES = RED_OFF + (NTH / 64) * 12;
  static_assert(M % 64 == 0 && 64 % T == 0 && pass_radix4(M, 0) == 4, "whole-wave rows, lines inside a wave, radix-4 first pass");
  static_assert(A_THREADS % 64 == 0 && A_THREADS < NTH && ROWD % 2 == 0, "whole waves per transform role");
};

// the unit roots of a line's slots: w[jt + q T] = w[jt] * e^{-2 pi i q / 8}  (T / nz = 1 / 8)
__device__ __forceinline__ cplx slot_root(cplx w, int q) {
  constexpr double h = 0.70710678118654752440;
  switch (q & 3) {
    case 0: return w;
    case 1: return cmake(h * (w.re + w.im), h * (w.im - w.re));    // w * (h - i h)
    case 2: return cmake(w.im, -w.re);                             // w * (-i)
    default: return cmake(h * (w.im - w.re), -h * (w.re + w.im));  // w * (-h - i h)
  }
}

// the phases of one line transform, separated by wave-local fences (all T threads of the line are lanes of one wave).
// The fences name the LDS address space: a plain wavefront fence also waits for the vector-memory counter, i.e. for the
// spectrum of the next plane that was requested just before the transform -- the prefetch would be waited for on the spot.
template <int M, int DIR, int PH>
__device__ __forceinline__ void line_phases(cplx* v, int jt, double* lds, const LdsMap& L, int line, const cplx* tw) {
  Line4<M>::template phase<DIR, PH>(v, jt, lds, L, line, tw);
  if constexpr (PH + 1 < Line4<M>::NPHASE) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    line_phases<M, DIR, PH + 1>(v, jt, lds, L, line, tw);
  }
}

// NL lines per thread at once (same jt, different line regions): the phases of all of them between two fences
template <int M, int DIR, int PH, int NL>
__device__ __forceinline__ void line_phases_n(cplx (*v)[4], int jt, double* lds, const LdsMap& L, const int* line, const cplx* tw) {
#pragma unroll
  for (int n = 0; n < NL; ++n) Line4<M>::template phase<DIR, PH>(v[n], jt, lds, L, line[n], tw);
  if constexpr (PH + 1 < Line4<M>::NPHASE) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    line_phases_n<M, DIR, PH + 1, NL>(v, jt, lds, L, line, tw);
  }
}

template <int M, int NR, bool SUMT, bool PHI2, int NT>
__global__ __launch_bounds__(NR * M) void k_uz_tile(Grid g, double beta, double gamma, FieldPtrs<3> u, FieldPtrs<2> mod,
                                                    FieldPtrs<3> fo, Vec6 E, double* partial, int nty, int LX, PhaseLin lin,
                                                    const cplx* twg, const cplx* wzg, cplx* dump) {
  using G = UzGeom<M, NR>;
  constexpr int TY = G::TY, NZS = G::NZS, T = G::T, ROWD = G::ROWD;
  constexpr int NS = SUMT ? 12 : 6;
  extern __shared__ __align__(16) double lds[];
  double* const Ub = lds + G::U_OFF;
  double* const Fb = lds + G::F_OFF;
  cplx* const tw = reinterpret_cast<cplx*>(lds + G::TW_OFF);
  cplx* const wzl = reinterpret_cast<cplx*>(lds + G::WZ_OFF);
  double* const red = lds + G::RED_OFF;
  const LdsMap LM = {1, ROWD, G::LS};   // a line's region: re plane, im plane

  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
  const int r = tid / M, li = tid % M, zs = wv % NZS;
  const int zprev = (zs + NZS - 1) % NZS, znext = (zs + 1) % NZS;
  int b = blockIdx.x;
  {
    const int nb = gridDim.x;
    if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;   // one XCD takes a contiguous run of tiles (halo rows meet in its L2)
  }
  const int ty = b % nty, tx = b / nty;
  if (tx * LX >= g.nx) {   // padding workgroup (grid rounded up to a multiple of 8)
    if (tid < NS) partial[(long)blockIdx.x * NS + tid] = 0.0;
    return;
  }
  const int j0 = min(ty * TY, g.ny - TY), x0 = tx * LX;
  const int nsteps = x0 + LX <= g.nx ? LX : g.nx - x0;
  auto wrap_y = [&](int jr) { return jr < 0 ? jr + g.ny : (jr >= g.ny ? jr - g.ny : jr); };
  auto plane = [&](int q) {   // element offset of x plane q (periodic; q in [-1, 2 nx))
    const int x = q < 0 ? q + g.xw_lo : (q >= g.nx ? q - g.xw_hi : q);
    return (long)x * g.nyzp;
  };
  // ---- stencil role
  const int jr = j0 - 1 + r;
  const bool own = r >= 1 && r <= TY && jr >= ty * TY;
  const long srow = (long)wrap_y(jr) * g.nzp + 2 * li;
  const double hx = g.hx, hy = g.hy, hz = g.hz;
  // bases at row r - 1 (see UzGeom): U images, tau exchange, tau edges; the F image at row r - 1 of the owned rows
  constexpr int UR = ROWD / 2, UC = NR * ROWD / 2;   // row / component strides of a U image in pairs
  const double2* const um0 = reinterpret_cast<const double2*>(Ub) + ((r - 1) * UR + li);
  double2* const tm = reinterpret_cast<double2*>(lds + G::TB_OFF) + ((r - 1) * M + li);
  double* const em = lds + G::EDGE_OFF + r * NZS;
  double* const fm = Fb + (r - 1) * ROWD + pad8(li);
  auto prev_y = [&](double v) { return dpp_move<0x138>(v); };   // lane i <- i-1
  auto next_x = [&](double v) { return dpp_move<0x130>(v); };   // lane i <- i+1
  // ---- c2r role: line = comp * NR + tile row
  const bool a_thread = tid < G::A_THREADS;
  const int a_line = a_thread ? tid / T : 0, a_jt = tid % T;
  const int a_comp = a_line / NR;
  const long a_row = (long)wrap_y(j0 - 1 + a_line % NR) * g.nzp;
  // ---- r2c role: line = comp * TY + (tile row - 1), LPR lines per round
  const int c_jt = tid % T;
  const int fft_jt = tid % T;
  auto mirror_lane = [&]() { return (l & ~(T - 1)) | ((T - fft_jt) & (T - 1)); };   // lane holding the mirrored slots of this line

  for (int i = tid; i < tw_total4(M); i += G::NTH) tw[i] = twg[i];
  if (tid < T) wzl[tid] = wzg[tid];

  cplx x[4];   // spectrum of the plane the c2r role transforms next
  double xny = 0.0;   // real part of its Nyquist bin (the imaginary part is ignored, so it is not even loaded: the compiler
                      // would reuse the dead half of a 16-byte load's destination at once and wait for the load to do so)
  auto a_load = [&](int q) {
    // (a per-thread index into the pointer array of the kernel arguments would be a global load + wait; select instead)
    const double* const ub = a_comp == 0 ? u.p[0] : (a_comp == 1 ? u.p[1] : u.p[2]);
    const cplx* in = reinterpret_cast<const cplx*>(ub + plane(q) + a_row);
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = cload_stream(&in[a_jt + s * T], NT);
    xny = (NT & 2) ? __builtin_nontemporal_load(&in[M].re) : in[M].re;   // every lane (one address per line): no branch around a load
  };
  // merge (c2r_merge, every coefficient read once: the mirrored one comes from the lane that loaded it, see C2RKernel),
  // inverse transform in the line's region, natural-order reals into the U image, request the spectrum of plane `pf`
  auto a_phase = [&](int img, bool prefetch, int pf) {
    cplx v[4];
    const cplx wz_jt = wzl[a_jt];
    const int mlane = mirror_lane();
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // first_index(jt, q) = jt + q T, so X[M - m] is slot 3 - q of lane T - jt (jt = 0: own slot 4 - q)
      cplx xm = cmake(__shfl(x[3 - q].re, mlane), __shfl(x[3 - q].im, mlane));
      if (a_jt == 0) xm = q == 0 ? cmake(xny, 0.0) : x[4 - (q ? q : 4)];
      cplx xk = x[q];
      if (a_jt == 0 && q == 0) { xk.im = 0.0; xm.im = 0.0; }   // FFTW's c2r ignores the imaginary parts of DC and Nyquist
      v[q] = c2r_merge(xk, xm, slot_root(wz_jt, q));
    }
    // The request for the next spectrum: right after the merge has consumed the old one, so that the loads have the whole
    // transform to land (the copy of the moduli at the end of the step waits for EVERY outstanding load).  Unconditional,
    // and every load of the loop in straight-line code: behind a branch the compiler's wait-count bookkeeping turns
    // conservative.  (Only the real part of the Nyquist bin is loaded: see xny.)
    (void)prefetch;
    __builtin_amdgcn_sched_barrier(0);
    if (!(FG_UZ_ABLATE & 2)) a_load(pf);
    __builtin_amdgcn_sched_barrier(0);
    double* const image = Ub + img * G::U_IMG;
    // (the padded exchange addresses of a line are loop invariants the compiler would park in ~40 registers -- and spill;
    // an opaque copy of the thread's line index makes it rebuild them, a few integer operations each)
    int jt = a_jt, line = a_line;
    asm volatile("" : "+v"(jt), "+v"(line));
    line_phases<M, +1, 0>(v, jt, image, LM, line, tw);
    double2* row = reinterpret_cast<double2*>(image + line * ROWD);
#pragma unroll
    for (int q = 0; q < 4; ++q) row[Line4<M>::last_index(jt, q)] = make_double2(v[q].re, v[q].im);
  };

  // ---- prologue: plane x0 - 1 in image 0, plane x0 in image 1, the spectrum of plane x0 + 1 requested
  if (a_thread) a_load(x0 - 1);
  __syncthreads();   // twiddles
  if (a_thread) {
    a_phase(0, true, x0);
    a_phase(1, true, x0 + 1);
  }
  double2 Ac = ld2(mod.p[0], plane(x0 - 1) + srow), Bc = Ac;
  if (!PHI2) Bc = ld2(mod.p[1], plane(x0 - 1) + srow);
  __syncthreads();

  double acc[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) acc[c] = 0.0;

  // The march, once per transform role: the c2r waves carry the spectrum of the next plane through the stencil phases, the
  // r2c waves the interleaved transforms of several lines -- as two instances of the loop neither pays for the other's
  // registers.  The role is uniform per wave (whole waves per role), and both instances execute the same three barriers
  // per step.
  auto march = [&](auto role) {
    double2 dx1 = make_double2(0.0, 0.0), dx2 = dx1;
    double2 t0m = dx1, t5m = dx1, t4m = dx1, part1 = dx1, part2 = dx1;
    for (int st = -1; st <= nsteps; ++st) {
      const int q = x0 + st;
      const int cur = (st + 1) & 1;   // image of plane q; the other one holds plane q + 1
      FG_UZ_MARK(0);
      // ---- B1
      const double2* const um = um0 + cur * (G::U_IMG / 2);          // plane q
      const double2* const un_ = um0 + (cur ^ 1) * (G::U_IMG / 2);   // plane q + 1
      double2 uc[3], un[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        uc[c] = um[c * UC + UR];
        un[c] = un_[c * UC + UR];
      }
      const double2 U0yb = um[0 * UC], U1yf = um[1 * UC + 2 * UR], U2yb = um[2 * UC];
      double U0zb = prev_y(uc[0].y), U1zb = prev_y(uc[1].y), U2zf = next_x(uc[2].x);
      if (l == 0) {   // wave edges: the neighbour pair lives in the adjacent wave of the same row (periodic in z)
        U0zb = um[0 * UC + UR + (zprev - zs) * 64 + 63].y;
        U1zb = um[1 * UC + UR + (zprev - zs) * 64 + 63].y;
      }
      if (l == 63) U2zf = um[2 * UC + UR + (znext - zs) * 64 - 63].x;
      // strain of the two voxels  (F:18632-18686)
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
      // polarisation  tau = (A - 2 mu0) eps + (B - lambda0) tr(eps) I
      const double ax = PHI2 ? lin.a0 + Ac.x * lin.da : Ac.x + beta, ay = PHI2 ? lin.a0 + Ac.y * lin.da : Ac.y + beta;
      const double bx = PHI2 ? lin.b0 + Ac.x * lin.db : Bc.x + gamma, by = PHI2 ? lin.b0 + Ac.y * lin.db : Bc.y + gamma;
      // the moduli of the next plane into the same registers, right after their last use (a second register set would
      // need a copy at the end of the step, and a copy of a register a load is still filling is a wait for that load)
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler would hoist the load above the last use -- into a second register set)
      Ac = ld2(mod.p[0], plane(q + 1) + srow);
      if (!PHI2) Bc = ld2(mod.p[1], plane(q + 1) + srow);
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
      // y neighbours of tau through LDS
      constexpr int TC = NR * M, EC = NR * NZS;   // component strides of the tau exchange and of its edge values
      tm[0 * TC + M] = t1;
      tm[1 * TC + M] = t5;
      tm[2 * TC + M] = t3;
      if (l == 63) em[0 * EC + zs] = t2.y;
      if (l == 0) {
        em[1 * EC + zs] = t3.x;
        em[2 * EC + zs] = t4.x;
      }
      FG_UZ_MARK(1);
      __syncthreads();
      FG_UZ_MARK(2);
      // ---- B2: divergence -> F image (f0 of this plane, f1 / f2 of the previous one)
      const double2 t1yb = tm[0 * TC], t5yf = tm[1 * TC + 2 * M], t3yf = tm[2 * TC + 2 * M];
      double t2zb = prev_y(t2.y), t3zf = next_x(t3.x), t4zf = next_x(t4.x);
      if (l == 0) t2zb = em[0 * EC + zprev];
      if (l == 63) {
        t3zf = em[1 * EC + znext];
        t4zf = em[2 * EC + znext];
      }
      if (r >= 1 && r <= TY) {
        // packed line point li of the row's line: z_li = f[2 li] + i f[2 li + 1]
        const cplx f0 = cmake((t0.x - t0m.x) * hx + (t5yf.x - t5.x) * hy + (t4.y - t4.x) * hz,
                              (t0.y - t0m.y) * hx + (t5yf.y - t5.y) * hy + (t4zf - t4.y) * hz);
        constexpr int FC = TY * ROWD, IM = G::LS;
        fm[0 * FC] = f0.re;
        fm[0 * FC + IM] = f0.im;
        fm[1 * FC] = (t5.x - t5m.x) * hx + part1.x;
        fm[1 * FC + IM] = (t5.y - t5m.y) * hx + part1.y;
        fm[2 * FC] = (t4.x - t4m.x) * hx + part2.x;
        fm[2 * FC + IM] = (t4.y - t4m.y) * hx + part2.y;
      }
      part1.x = (t1.x - t1yb.x) * hy + (t3.y - t3.x) * hz;
      part1.y = (t1.y - t1yb.y) * hy + (t3zf - t3.y) * hz;
      part2.x = (t3yf.x - t3.x) * hy + (t2.x - t2zb) * hz;
      part2.y = (t3yf.y - t3.y) * hy + (t2.y - t2.x) * hz;
      // advance one plane
      t0m = t0; t5m = t5; t4m = t4;
      dx1.x = un[1].x - uc[1].x; dx1.y = un[1].y - uc[1].y;
      dx2.x = un[2].x - uc[2].x; dx2.y = un[2].y - uc[2].y;
      FG_UZ_MARK(3);
      __syncthreads();
      FG_UZ_MARK(4);
      // ---- c2r of plane q + 2 over the image of plane q (read by the B1 of the next step as ITS plane q + 1; the last
      //      step needs none), beside the r2c of the finished lines: f0 of plane q (steps 0 .. nsteps-1), f1 / f2 of plane
      //      q - 1 (steps 1 .. nsteps)
      if constexpr (decltype(role)::value == 0) {
        if (st <= nsteps - 2) a_phase(cur, st <= nsteps - 3, q + 3);
      } else {
        // the NRND lines of this thread TOGETHER (interleaved instruction streams, shared fences): the r2c waves carry more
        // lines than the c2r waves, and run one after the other each line's chain of LDS round trips would leave the SIMD to
        // the last of them alone
        constexpr int NRND = G::NRND;
        constexpr int RL = pass_radix4(M, num_passes4(M) - 1), GG = 4 / RL;
        cplx v[NRND][4];
        int line[NRND], jt = c_jt;
        bool live[NRND];
        asm volatile("" : "+v"(jt));   // see a_phase
#pragma unroll
        for (int n = 0; n < NRND; ++n) {
          line[n] = (tid - G::C0) / T + n * G::LPR;
          asm volatile("" : "+v"(line[n]));
          live[n] = line[n] < G::NLC && (line[n] / TY == 0 ? inside : st >= 1);
          if (line[n] >= G::NLC) line[n] = G::NLC;   // the spare region (shared garbage; an owned line must not be touched)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) v[n][s4] = lds_get(Fb, LM, Line4<M>::first_index(jt, s4), line[n]);
        }
        FG_UZ_MARK(8);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        line_phases_n<M, -1, 0, NRND>(v, jt, Fb, LM, line, tw);
        FG_UZ_MARK(9);
        const cplx wz_jt = wzl[c_jt];
        const int mlane = mirror_lane();
#pragma unroll
        for (int n = 0; n < NRND; ++n) {
          // slot s' = (b, rr) of the last pass (radix RL) holds Z[jt + s T], s = b + rr (4 / RL); Z[M - k] is slot 3 - s of
          // lane T - jt of this line (jt = 0: the own slot 4 - s, Z[M] := Z[0])  -- see R2CKernel
          const int c_comp = min(line[n] / TY, 2), c_r = 1 + line[n] % TY;
          // Every store is issued, those of lines that are not due (ends of the march, rows the neighbouring tile owns) to
          // a dump slot of the thread: stores behind a branch would make the compiler's wait for the NEXT load of the loop
          // (the moduli) a wait for all of them (the in-order counter cannot skip stores that may not have been issued).
          const bool c_store = !(FG_UZ_ABLATE & 1) && live[n] && j0 - 1 + c_r >= ty * TY;
          double* const fb = c_comp == 0 ? fo.p[0] : (c_comp == 1 ? fo.p[1] : fo.p[2]);   // (no indexed kernel-argument load)
          cplx* const row = reinterpret_cast<cplx*>(fb + plane(c_comp == 0 ? q : q - 1) + (long)wrap_y(j0 - 1 + c_r) * g.nzp);
          cplx* const sink = dump + (blockIdx.x % kUzDumpGroups) * G::NTH + tid;
          cplx first = cmake(0.0, 0.0);
#pragma unroll
          for (int s8 = 0; s8 < 4; ++s8) {
            const int s = s8 / RL + (s8 % RL) * GG;
            const int sm = 3 - s, qm = (sm % GG) * RL + sm / GG;
            const int s0 = (4 - s) % 4, q0 = (s0 % GG) * RL + s0 / GG;
            cplx zm = cmake(__shfl(v[n][qm].re, mlane), __shfl(v[n][qm].im, mlane));
            if (c_jt == 0) zm = v[n][q0];
            const int k = c_jt + s * T;
            const cplx xk = r2c_split(v[n][s8], zm, slot_root(wz_jt, s));
            if (s8 == 0) first = xk;   // (slot 0 is s = 0: k = jt)
            cstore_stream(c_store ? &row[k] : sink, xk, NT);
          }
          // k = M: Z[M] := Z[0], w = -1, by the line's first thread; the others store their own X[jt] once more
          const cplx xM = r2c_split(v[n][0], v[n][0], cmake(-1.0, 0.0));
          cstore_stream(c_store ? &row[c_jt == 0 ? M : c_jt] : sink, c_jt == 0 ? xM : first, NT);
        }
      }
      FG_UZ_MARK(5);
      __syncthreads();
      FG_UZ_MARK(6);
    }
  };
  if (a_thread) march(std::integral_constant<int, 0>{});
  else march(std::integral_constant<int, 1>{});

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
  if (tid < NS) {
    double a = 0.0;
    for (int w = 0; w < G::NTH / 64; ++w) a += red[w * NS + tid];
    partial[(long)blockIdx.x * NS + tid] = a;
  }
}

inline int uz_march_length(int nx, long tiles, int cus) {   // as k_u_tile's: a march of LX planes costs LX + 3 steps
  int best = nx < 4 ? nx : 4;
  long best_cost = -1;
  for (int lx = 4; lx <= nx && lx <= 64; ++lx) {
    const long groups = tiles * ((nx + lx - 1) / lx);
    const long cost = ((groups + cus - 1) / cus) * (lx + 3);
    if (best_cost < 0 || cost <= best_cost) best = lx, best_cost = cost;
  }
  return best;
}

template <int M, int NR, bool SUMT, bool PHI2>
void launch_uz_t(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& u, const FieldPtrs<2>& mod,
                 const FieldPtrs<3>& f, const Vec6& E, double* partial, double* sumsq6, hipStream_t s, const PhaseLin& lin,
                 const cplx* tw_z, const cplx* w_z) {
  using G = UzGeom<M, NR>;
  const int nty = (g.ny + G::TY - 1) / G::TY;
  static const int lx_env = getenv("FG_UZ_LX") ? atoi(getenv("FG_UZ_LX")) : 0;
  int LX = lx_env > 0 ? lx_env : uz_march_length(g.nx, nty, device_cu_count());
  if (LX > g.nx) LX = g.nx;
  const int ntx = (g.nx + LX - 1) / LX;
  int nb = nty * ntx;
  if (nb >= 8) nb = ((nb + 7) / 8) * 8;
  const size_t lds = (size_t)G::LDS_DOUBLES * sizeof(double);
  static cplx* dump[kMaxDevices] = {};   // where the stores of lines that are not due go (one slot per thread; never read)
  static PerDeviceOnce configured;
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipMalloc(&dump[current_device()], (size_t)kUzDumpGroups * G::NTH * sizeof(cplx)));
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_uz_tile<M, NR, SUMT, PHI2, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_uz_tile<M, NR, SUMT, PHI2, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // streaming loads (2) and stores (1) of the spectra when the fields exceed the Infinity Cache, like the FFT passes
  static const int nt_env = getenv("FG_UZ_NT") ? atoi(getenv("FG_UZ_NT")) : -1;
  const bool nt = nt_env >= 0 ? nt_env != 0 : 3.0 * (double)g.n * sizeof(double) > 256.0 * 1024 * 1024;
  if (nt)
    hipLaunchKernelGGL((k_uz_tile<M, NR, SUMT, PHI2, 3>), dim3(nb), dim3(G::NTH), lds, s, g, -2 * mu_0, -lambda_0, u, mod, f, E,
                       partial, nty, LX, lin, tw_z, w_z, dump[current_device()]);
  else
    hipLaunchKernelGGL((k_uz_tile<M, NR, SUMT, PHI2, 0>), dim3(nb), dim3(G::NTH), lds, s, g, -2 * mu_0, -lambda_0, u, mod, f, E,
                       partial, nty, LX, lin, tw_z, w_z, dump[current_device()]);
  FG_HIP_CHECK(hipGetLastError());
  fold_sum(partial, nb, SUMT ? 12 : 6, sumsq6, s);
  FG_HIP_CHECK(hipGetLastError());
}

}  // namespace

bool uz_tile_supported(const Grid& g) {
  const int M = g.nz / 2;
  return g.nz % 2 == 0 && (M == 64 || M == 128) && g.nzc >= M + 1 && g.ny >= 16 && g.nx >= 4;
}

void launch_uz_tile(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& uhat, const FieldPtrs<2>& mod,
                    const FieldPtrs<3>& fhat, const Vec6& E, double* partial, double* sumsq6, hipStream_t s, bool sum_tau,
                    const PhaseTable* two_phase, const cplx* tw_z, const cplx* w_z) {
  PhaseLin lin = {0, 0, 0, 0};
  if (two_phase)
    lin = PhaseLin{2 * two_phase->mu[0] - 2 * mu_0, 2 * (two_phase->mu[1] - two_phase->mu[0]), two_phase->lambda[0] - lambda_0,
                   two_phase->lambda[1] - two_phase->lambda[0]};
  const int M = g.nz / 2;
#define FG_UZ(MM, RR)                                                                                                     \
  do {                                                                                                                    \
    if (two_phase) {                                                                                                      \
      if (sum_tau) launch_uz_t<MM, RR, true, true>(g, mu_0, lambda_0, uhat, mod, fhat, E, partial, sumsq6, s, lin, tw_z, w_z); \
      else launch_uz_t<MM, RR, false, true>(g, mu_0, lambda_0, uhat, mod, fhat, E, partial, sumsq6, s, lin, tw_z, w_z);        \
    } else {                                                                                                              \
      if (sum_tau) launch_uz_t<MM, RR, true, false>(g, mu_0, lambda_0, uhat, mod, fhat, E, partial, sumsq6, s, lin, tw_z, w_z); \
      else launch_uz_t<MM, RR, false, false>(g, mu_0, lambda_0, uhat, mod, fhat, E, partial, sumsq6, s, lin, tw_z, w_z);        \
    }                                                                                                                     \
  } while (0)
  static const int rows_env = getenv("FG_UZ_ROWS") ? atoi(getenv("FG_UZ_ROWS")) : 0;
  if (M == 128) {
    if (rows_env == 8) FG_UZ(128, 8);
    else FG_UZ(128, 6);
  } else if (M == 64) {
    if (rows_env == 16) FG_UZ(64, 16);
    else if (rows_env == 8) FG_UZ(64, 8);
    else FG_UZ(64, 12);
  } else {
    throw std::runtime_error("launch_uz_tile: unsupported nz");
  }
#undef FG_UZ
}

}  // namespace fg
