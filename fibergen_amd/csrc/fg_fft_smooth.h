// Transform passes for lengths that are NOT a power of two (nor 3, 5, 7, 9 times one): N = a product of the primes
// 2, 3, 5, 7, 11, 13 -- the decimal grid sizes users of the reference pick (100, 200, 300, 400, 500 ...; FFTW takes any).
//
// Stockham autosort on a tile of lines held in LDS, in the style of the power-of-two kernels (a thread = one butterfly held in
// registers): a workgroup loads `lines` whole lines (adjacent columns for the strided y / x passes -- one 128-byte segment
// per line point for 8 columns --, rows for the z passes) into ONE LDS image and runs a few passes with LARGE radices R <= 32
// (N = 500 is 25 x 20, 300 is 25 x 12, 1000 is 10 x 10 x 10): in a pass thread (j, t) reads the points j + r N/R of line t,
// multiplies by w^{r (j mod Ns) N/(Ns R)} (Ns = product of the radices before: none in the first pass), transforms the R
// values in registers -- a composite butterfly built from the prime / 4 / 8 butterflies with constant twiddles --, and,
// behind a barrier, writes them to the points (j / Ns) Ns R + (j mod Ns) + r Ns of the same image: natural order after the
// last pass.  The planner picks the radices so that every pass has at most one butterfly per thread (N lines / R <= threads).
// One read and one write of the data per axis, like the power-of-two kernels (which stay the faster ones for their lengths).
// The per-thread code is FG_HD: tests/emulate runs it on the host against numpy.
#pragma once

#include "fg_fft_core.h"
#include "fg_fft_roots.h"
#include "fg_stage_math.h"

namespace fg {
namespace fft {

constexpr int kSmoothMaxPasses = 4;
constexpr int kSmoothMaxRadix = 32;
constexpr size_t kSmoothLdsMax = 156 * 1024;

struct SmoothPlan {
  int n = 0;         // line length (z passes: nz / 2); 0 = no plan
  int npass = 0;
  int fac[kSmoothMaxPasses] = {};
  int lines = 0;     // lines per tile (strided passes: columns)
  int threads = 0;   // 256 or 1024
  int cap = 20;      // values a thread holds in a pass (smooth_rounds: several butterflies of a small radix); 0: one butterfly
  int joint = 0;     // fused x pass: the components (1 or 3) whose columns share ONE image [p][joint][columns] (lines = joint * columns,
                     // k_smooth_xjoint); 0: one image and one pass loop per component (k_smooth_xfused)
  int rmax() const { int m = 0; for (int i = 0; i < npass; ++i) m = fac[i] > m ? fac[i] : m; return m; }
};

constexpr bool smooth_is_base(int r) { return r == 2 || r == 4 || r == 8 || r == 3 || r == 5 || r == 7 || r == 11 || r == 13; }
constexpr int smooth_first_factor(int r) {
  for (int f : {13, 11, 7, 5, 3, 8, 4, 2})
    if (r % f == 0) return f;
  return 0;
}
constexpr bool smooth_radix_ok(int r) {   // 2 ... 32 with prime factors up to 13
  if (r < 2 || r > kSmoothMaxRadix) return false;
  for (int f : {2, 3, 5, 7, 11, 13})
    while (r % f == 0) r /= f;
  return r == 1;
}

// Natural-order DFT of an odd prime length: with a_m = x_m + x_{R-m}, b_m = x_m - x_{R-m} (m = 1 .. h = (R - 1) / 2)
//   X_k, X_{R-k} = x_0 + sum_m cos(2 pi m k / R) a_m  +-  DIR i sum_m sin(2 pi m k / R) b_m
template <int R, int DIR>
FG_HD void dft_odd(cplx* v) {
  constexpr int H = (R - 1) / 2;
  cplx a[H], b[H];
#pragma unroll
  for (int m = 0; m < H; ++m) {
    a[m] = cadd(v[m + 1], v[R - 1 - m]);
    b[m] = csub(v[m + 1], v[R - 1 - m]);
  }
  const cplx x0 = v[0];
  cplx sum = x0;
#pragma unroll
  for (int m = 0; m < H; ++m) sum = cadd(sum, a[m]);
  v[0] = sum;
#pragma unroll
  for (int k = 1; k <= H; ++k) {
    double cr = x0.re, ci = x0.im, sr = 0.0, si = 0.0;
#pragma unroll
    for (int m = 1; m <= H; ++m) {
      const int j = (m * k) % R;   // compile-time after unrolling
      const double cj = root_cos<R>(j), sj = root_sin<R>(j);
      cr += cj * a[m - 1].re;
      ci += cj * a[m - 1].im;
      sr += sj * b[m - 1].re;
      si += sj * b[m - 1].im;
    }
    // DIR i (sr + i si) = DIR (-si + i sr)
    v[k] = cmake(cr - DIR * si, ci + DIR * sr);
    v[R - k] = cmake(cr + DIR * si, ci - DIR * sr);
  }
}

// Natural-order DFT of R values in registers, R any product of the base radices: Cooley-Tukey with R = R1 R2, input index
// n = R2 n1 + n2, output index k = k1 + R1 k2: R2 transforms of length R1 over n1, the constant twiddles w_R^{n2 k1}, R1
// transforms of length R2 over n2.
template <int R, int DIR>
FG_HD void dft_reg(cplx* v) {
  if constexpr (R == 2) dft2<DIR>(v);
  else if constexpr (R == 4) dft4<DIR>(v);
  else if constexpr (R == 8) dft8<DIR>(v);
  else if constexpr (smooth_is_base(R)) dft_odd<R, DIR>(v);
  else {
    constexpr int R1 = smooth_first_factor(R), R2 = R / R1;
    cplx a[R];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
      cplx col[R1];
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) col[n1] = v[R2 * n1 + n2];
      dft_reg<R1, DIR>(col);
#pragma unroll
      for (int k1 = 0; k1 < R1; ++k1) {
        const int m = (n2 * k1) % R;   // compile-time after unrolling
        a[k1 * R2 + n2] = m == 0 ? col[k1] : cmul(col[k1], cmake(root_cos<R>(m), DIR * root_sin<R>(m)));
      }
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
      dft_reg<R2, DIR>(&a[k1 * R2]);
#pragma unroll
      for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = a[k1 * R2 + k2];
    }
  }
}

// Addressing of a tile in LDS: point p of line t sits at p * sp + t * sc.
//   strided passes: sp = lines, sc = 1 (the columns of a line point are adjacent, as in memory); threads run over t fastest
//   z passes:       sp = 1, sc = line pitch (a row is contiguous, as in memory);                  threads run over j fastest
// idx / d without an integer division: exact for every 0 <= idx < 2^31 (the double's error is far below 1 / (2 d))
FG_HD int smooth_div(int idx, double inv_d) { return (int)(((double)idx + 0.5) * inv_d); }
// the same in single precision for the small operands of the passes (thread and butterfly numbers): exact while a < 2^21
// (error of the product <= 3 a / d 2^-24 against a distance of 0.5 / d to the next integer)
FG_HD int smooth_divf(int a, float inv_d) { return (int)(((float)a + 0.5f) * inv_d); }

struct SmoothMap {
  int sp, sc, lines;
  bool jfast;
};

// First half of a pass for thread `tid`: the R inputs of its butterfly, twiddled and transformed -> v.  Returns whether
// the thread owns a butterfly.  w: entry k * wscale = e^{-2 pi i k / N}.
// FDIV: index arithmetic in 32 bits and without integer divisions by run-time values (a single-precision reciprocal is exact
// here and costs 4 instructions where the division sequence costs ~35 -- seven of them per butterfly and pass; SQ_INSTS_VALU of
// the 400^3 passes: 2.0-2.4 x the power-of-two kernels' per voxel).  y / z passes 3-8 % faster with it; the fused x pass, which
// sits at its 256-VGPR cap, 10 % SLOWER (the compiler then keeps all R addresses of a butterfly in registers: 26 -> 107 spilled):
// it keeps the plain form (its tile width is a compile-time constant, two of the seven divisions fold anyway).
template <int R, int DIR, bool FDIV = true>
FG_HD bool smooth_pass_read(const cplx* img, int N, int Ns, const SmoothMap& L, const cplx* w, int wscale, int tid, cplx* v) {
  const int nb = N / R;
  if (tid >= nb * L.lines) return false;
  int t, j, k = 0;
  if (FDIV) {
    const int d = L.jfast ? nb : L.lines;
    const int q = smooth_divf(tid, 1.0f / (float)d), rem = tid - q * d;
    t = L.jfast ? q : rem, j = L.jfast ? rem : q;
    const cplx* src = img + t * L.sc;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[(j + r * nb) * L.sp];
    if (Ns > 1) {
      const float inv_ns = 1.0f / (float)Ns;
      k = (j - smooth_divf(j, inv_ns) * Ns) * smooth_divf(nb, inv_ns) * wscale;
    }
  } else {
    t = L.jfast ? tid / nb : tid % L.lines;
    j = L.jfast ? tid % nb : tid / L.lines;
    const cplx* src = img + (long)t * L.sc;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[(long)(j + r * nb) * L.sp];
    if (Ns > 1) k = (j % Ns) * (N / (Ns * R)) * wscale;
  }
  if (Ns > 1) {
    // w^{r m}, m = (j mod Ns) N / (Ns R): the first power from the table, the others as products of two lower ones (error
    // growth ~ log2 R roundings)
    cplx p[R];
    p[1] = w[k];
    if (DIR > 0) p[1] = cconj(p[1]);
    v[1] = cmul(v[1], p[1]);
#pragma unroll
    for (int r = 2; r < R; ++r) {
      p[r] = cmul(p[r / 2], p[r - r / 2]);
      v[r] = cmul(v[r], p[r]);
    }
  }
  dft_reg<R, DIR>(v);
  return true;
}

// Second half (behind a barrier: every thread has read its inputs): the outputs to their Stockham positions
template <int R, bool FDIV = true>
FG_HD void smooth_pass_write(cplx* img, int N, int Ns, const SmoothMap& L, int tid, const cplx* v) {
  const int nb = N / R;
  if (FDIV) {
    const int d = L.jfast ? nb : L.lines;
    const int q = smooth_divf(tid, 1.0f / (float)d), rem = tid - q * d;
    const int t = L.jfast ? q : rem, j = L.jfast ? rem : q;
    const int jh = Ns > 1 ? smooth_divf(j, 1.0f / (float)Ns) : j, jl = j - jh * Ns;
    cplx* dst = img + t * L.sc + (jh * Ns * R + jl) * L.sp;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[r * Ns * L.sp] = v[r];
  } else {
    const int t = L.jfast ? tid / nb : tid % L.lines;
    const int j = L.jfast ? tid % nb : tid / L.lines;
    cplx* dst = img + (long)t * L.sc + (long)((j / Ns) * Ns * R + j % Ns) * L.sp;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[(long)r * Ns * L.sp] = v[r];
  }
}

// A thread may own several butterflies of a SMALL radix in one pass (their values stay in registers across the barrier like
// one large butterfly's): up to smooth_rounds(R) of them, R * rounds <= 20 values.  That lets lengths such as 200 = 8 x 5 x 5
// or 500 = 10 x 10 x 5 run with radices <= 16 -- half the registers and twice the resident workgroups of the R <= 32 kernels.
// (cap = the values a thread may hold: 20 in the kernels built for radices <= 16, 32 in the joint fused x pass of the R <= 32 class)
constexpr int smooth_rounds(int r, int cap = 20) { return 2 * r > cap ? 1 : (cap / r > 8 ? 8 : cap / r); }

// ---- planner: lines per tile, threads per workgroup and the radices of the passes
// radices of n in [lo, hi], at most kSmoothMaxPasses of them: fewest passes first, then the smallest largest radix; returned
// largest first (the first pass has no twiddles)
struct SmoothSearch {
  int lo, hi, best[kSmoothMaxPasses], cur[kSmoothMaxPasses], nbest = 0;
  long n = 0, lines = 0, threads = 0;   // radix r fits a pass when its n / r butterflies per line need <= smooth_rounds(r) rounds
  int cap = 20;                         // values per thread (smooth_rounds); 0: one butterfly per thread and pass only
  bool fits(int r) const { return threads == 0 || (n / r) * lines <= threads * (long)(cap ? smooth_rounds(r, cap) : 1); }
  void go(int rest, int depth, int maxr) {
    if (rest == 1 && depth > 0) {
      bool better = !nbest || depth < nbest;
      if (!better && depth == nbest) better = cur[0] < best[0];   // (non-increasing lists: entry 0 is the largest radix)
      if (better) {
        for (int i = 0; i < depth; ++i) best[i] = cur[i];
        nbest = depth;
      }
      return;
    }
    if (depth == kSmoothMaxPasses || (nbest && depth + 1 > nbest)) return;
    for (int r = maxr < hi ? maxr : hi; r >= lo && r >= 2; --r) {   // non-increasing: every multiset once
      if (rest % r || !smooth_radix_ok(r) || !fits(r)) continue;
      cur[depth] = r;
      go(rest / r, depth + 1, r);
    }
  }
};

inline bool smooth_factor(int n, int lines, int threads, int hi, int* fac, int* npass, int cap = 20) {
  SmoothSearch s;
  s.lo = 2;
  s.hi = hi;
  s.n = n, s.lines = lines, s.threads = threads;
  s.cap = cap;
  s.go(n, 0, hi);
  if (!s.nbest) return false;
  for (int i = 0; i < s.nbest; ++i) fac[i] = s.best[i];
  *npass = s.nbest;
  return true;
}

FG_HD int smooth_z_pitch(int M) { return M + 1 + ((M + 1) % 2 == 0 ? 1 : 0); }   // odd: rows start on different banks
// thread map of the z passes: lanes across the tile's ROWS (row pitch odd: conflict-free reads and writes in every pass).  With
// lanes along a row (the first form) the first pass writes its outputs R elements apart -- 8-way bank conflicts for R = 8:
// SQ_LDS_BANK_CONFLICT 86 M of 199 M LDS cycles in the 400^3 z passes
FG_HD SmoothMap smooth_z_map(int M, int lines) { return SmoothMap{1, smooth_z_pitch(M), lines, false}; }

inline bool smooth_try(int n, int lines, int threads, int hi, SmoothPlan* p, int cap = 20) {
  int fac[kSmoothMaxPasses], np = 0;
  if (!smooth_factor(n, lines, threads, hi, fac, &np, cap)) return false;
  p->n = n;
  p->npass = np;
  for (int i = 0; i < np; ++i) p->fac[i] = fac[i];
  p->lines = lines;
  p->threads = threads;
  p->cap = cap;
  return true;
}

// strided pass over lines of n points: 8 columns (one 128-byte segment per line point) where the image fits, 256 threads where
// radices up to 32 cover n * columns / 256 points per thread, else 1024 threads (radices up to 16)
inline bool smooth_plan_strided(int n, SmoothPlan* p) {
  *p = SmoothPlan();
  if (n < 2) return false;
  for (int cols : {32, 16, 8, 4, 2}) {
    // short lines take 16 or 32 columns (two / four 128-byte segments per line point) while the image stays <= 64 KB: as for the
    // z passes, the bytes a workgroup has in flight are the lever; with radices <= 16 only (the wide tiles are for small N).
    // 8 / wide alternating in one job: 100^3 4 190 -> 4 600 it/s, 120^3 +3 %, 200^3 +5.5 %, 240^3 +4 %, 250^3 +2.5 %
    if (cols > 8 && ((size_t)n * cols * sizeof(cplx) > (size_t)64 * 1024 || !smooth_try(n, cols, 256, 16, p))) continue;
    if (cols > 8) return true;
    if ((size_t)n * cols * sizeof(cplx) > kSmoothLdsMax) continue;
    // (256 threads, radices <= 16): half the registers, twice the resident workgroups; then radices <= 32; then 1024 threads
    if (smooth_try(n, cols, 256, 16, p) || smooth_try(n, cols, 256, kSmoothMaxRadix, p) || smooth_try(n, cols, 1024, 16, p)) return true;
  }
  return false;
}

// z pass over rows of M = nz / 2 complex points: as many rows per tile (a power of two <= 64) as keep the image at <= 64 KB
// (two workgroups per CU), at least one.  Measured: 48 KB -> 64 KB (16 instead of 8 rows of 200 points) 400^3 z passes
// 1 300 / 1 180 -> 960 / 890 us; 78 KB the same as 64; the point slots padded p + p / 8 against LDS bank conflicts: nothing.
inline bool smooth_plan_z(int M, SmoothPlan* p) {
  *p = SmoothPlan();
  if (M < 2) return false;
  const size_t line = (size_t)smooth_z_pitch(M) * sizeof(cplx);
  for (int round = 0; round < 2; ++round)
    for (int lines : {64, 32, 16, 8, 4, 2, 1}) {
      if (line * lines > (round == 0 ? (size_t)64 * 1024 : kSmoothLdsMax)) continue;
      if (smooth_try(M, lines, 256, 16, p) || smooth_try(M, lines, 256, kSmoothMaxRadix, p) || smooth_try(M, lines, 1024, 16, p)) return true;
    }
  return false;
}

// fused x pass (x transform, Green operator, inverse x transform on `ncomp` components of a tile): ncomp images in LDS.
// 8-column tiles only: measured against the three separate kernels, 100^3 +9 %, 120^3 +13 %, 300^3 +11 %, 400^3 +14 %, but with
// the 4-column tiles three components of 480 / 500 points need, 480^3 -1.4 %, 500^3 -2.3 % (half-line segments)
inline bool smooth_plan_xfused(int n, int ncomp, SmoothPlan* p, bool joint = true) {
  *p = SmoothPlan();
  if (n < 2) return false;
  for (int cols : {16, 8}) {
    if ((size_t)ncomp * n * cols * sizeof(cplx) > (cols > 8 ? (size_t)80 * 1024 : kSmoothLdsMax)) continue;   // 16 columns: short lines (100^3: fused x pass 46 -> 33 us; 120^3 at 92 KB: 57 -> 61)
    // this kernel holds the forward and the inverse butterflies and takes 256 registers in either class (the pass sets as
    // non-inlined calls: 200 VGPRs, and 1.7-2 x slower): the plan with the fewest passes wins, not the one with small radices
    // (and one butterfly per thread and pass: with the several-rounds code for small radices the kernel measured 100^3
    // 48 -> 68 us, 120^3 58 -> 90 us)
    SmoothPlan a, b;
    const bool ha = smooth_try(n, cols, 256, 16, &a, 0), hb = smooth_try(n, cols, 256, kSmoothMaxRadix, &b, 0);
    SmoothPlan best;
    if (ha && (!hb || a.npass <= b.npass)) best = a;
    else if (hb) best = b;
    else smooth_try(n, cols, 1024, 16, &best, 0);
    // joint image of the ncomp components (every pass runs ONCE, over ncomp x the butterflies: 3 x the threads at work and a
    // third of the barriers of the one-image-per-component form), where that takes no more passes
    if (joint) {
      SmoothPlan j, cand;
      // kernels built for (256 threads, radices <= 16), (256, <= 20), (512, <= 20) with 20 values per thread, (256, <= 32) with 32:
      // fewest passes, then the first of this list (registers follow the largest radix a kernel is built for)
      const int jt[4] = {256, 256, 512, 256}, jr[4] = {16, 20, 20, kSmoothMaxRadix}, jc[4] = {20, 20, 20, 32};
      for (int k = 0; k < 4; ++k) {
        if (k == 3 && j.n && ncomp == 3) break;   // three components: the R <= 32 form (one 256-thread workgroup at 512 VGPRs in its class kernel) only where no other fits
        if (smooth_try(n, ncomp * cols, jt[k], jr[k], &cand, jc[k]) && (!j.n || cand.npass < j.npass)) j = cand;
      }
      // (one pass more than the per-component plan where that one needs radices > 20: its kernels are built for R <= 32 and take
      // 512 VGPRs; 384 = 16 x 12 x 2 on the joint image against 24 x 16 per component)
      if (j.n && (!best.n || j.npass <= best.npass || (best.rmax() > 20 && j.rmax() <= 20 && j.npass <= best.npass + 1))) {
        j.joint = ncomp;
        best = j;
      }
    }
    if (best.n) { *p = best; return true; }
  }
  // longer lines (420 ... 800 points): a joint image of 4-column tiles (64-byte segments) in the 512-thread kernel.  As one image
  // per component in the R <= 32 kernels (512 VGPRs, one wave per SIMD) this form lost to three separate kernels (480^3 -1.4 %,
  // 500^3 -2.3 %); with 8 waves at 256 VGPRs it wins
  if (joint && ncomp == 3 && (size_t)ncomp * n * 4 * sizeof(cplx) <= kSmoothLdsMax && smooth_try(n, ncomp * 4, 512, 20, p, 20)) {
    p->joint = ncomp;
    return true;
  }
  return false;
}

struct SmoothArgs {
  cplx* data;          // component base
  long ls, os;         // line stride / outer stride (complex elements)
  int ncols, tiles_per_outer;
  double scale;
  const cplx* w;       // e^{-2 pi i k / n}, k < n
  int nt;
  SmoothPlan plan;
};

// fused x pass: columns are (ky, kz) pairs, col = jj * nzc + kk of the [x][ny][nzc] spectrum (jj0: first ky of a y-slab)
struct SmoothXArgs {
  SmoothArgs base;       // data = component 0
  long comp_stride;      // complex elements
  int ncomp;             // 3: elastic Green operator, 1: scalar c10 / |k|^2
  int nzc, nzf, jj0;
  double c10, c20;
  const double* kpm[3];
  const cplx* kp[3];
};

struct SmoothZArgs {
  double* data;        // component base (padded real rows / complex rows)
  long nrows;
  int nzp;
  const cplx* w;       // e^{-2 pi i k / nz}, k < nz  (its even entries are the roots of nz / 2)
  int nt;
  SmoothPlan plan;     // of M = nz / 2 (odd = 0) or of nz (odd = 1)
  int odd = 0;         // 1: nz is odd -- no packed-real trick: the row as nz complex points with zero imaginary parts
};

// ---- tile phases (between two workgroup barriers each; thread `tid` of `nthreads`)
// A thread moves its elements in batches of B: ALL global loads of a batch are issued before the first value is used -- a loop
// of load -> LDS store -> next load waits for the memory latency once per element (the first form of these kernels: 25 us per
// 500-point tile, most of it in such loops).  B = 16 with 256 threads (64 KB in flight per workgroup), 8 with 1024.
// the batched loads with the streaming flag decided ONCE per phase (a uniform branch around two straight-line batches), not per
// load: a flag test in front of every load puts each in a branch of its own (what round 5 found in the fused x pass)
template <bool NTL>
FG_HD cplx smooth_cload(const cplx* p) { return cload_stream(p, NTL ? 2 : 0); }


// strided pass: the tile = columns [col0, col0 + C) of outer index o, image [p][C]
template <int C, int B, bool NTL>
FG_HD void smooth_strided_load_impl(const SmoothArgs& a, int block, int tid, int nthreads, cplx* img) {
  const int o = block / a.tiles_per_outer, col0 = (block % a.tiles_per_outer) * C;
  const long base = (long)o * a.os + col0;
  const int total = a.plan.n * C;
  for (int i0 = tid; i0 < total; i0 += B * nthreads) {
    cplx v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int p = idx / C, t = idx % C;
      v[i] = idx < total && col0 + t < a.ncols ? smooth_cload<NTL>(&a.data[base + (long)p * a.ls + t]) : cmake(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      if (idx < total) img[idx] = v[i];
    }
  }
}

template <int C, int B>
FG_HD void smooth_strided_load(const SmoothArgs& a, int block, int tid, int nthreads, cplx* img) {
  if (a.nt & 2) smooth_strided_load_impl<C, B, true>(a, block, tid, nthreads, img);
  else smooth_strided_load_impl<C, B, false>(a, block, tid, nthreads, img);
}

template <int C>
FG_HD void smooth_strided_store(const SmoothArgs& a, int block, int tid, int nthreads, const cplx* img) {
  const int o = block / a.tiles_per_outer, col0 = (block % a.tiles_per_outer) * C;
  const long base = (long)o * a.os + col0;
  for (int idx = tid; idx < a.plan.n * C; idx += nthreads) {
    const int p = idx / C, t = idx % C;
    if (col0 + t < a.ncols) cstore_stream(&a.data[base + (long)p * a.ls + t], cscale(a.scale, img[idx]), a.nt);
  }
}

// fused x pass: G0OperatorFourierStaggeredGeneral F:19834-19927 (scalar modes: G0OperatorFourierStaggeredHeat F:19758-19823)
// on the transformed images [kx][C] of the tile's components, with the 1/N of fftVector (F:18501-18506) folded in
template <int C, int NC>
FG_HD void smooth_x_green(const SmoothXArgs& a, int block, int tid, int nthreads, cplx* img) {
  const int n = a.base.plan.n, col0 = (block % a.base.tiles_per_outer) * C;
  const long comp = (long)n * C;
  for (int idx = tid; idx < n * C; idx += nthreads) {
    const int kx = idx / C, t = idx % C, col = col0 + t;
    if (col >= a.base.ncols) continue;
    const int jl = smooth_div(col, 1.0 / (double)a.nzc), kk = col - jl * a.nzc, jj = a.jj0 + jl;
    if (kk >= a.nzf) continue;   // row padding
    const bool zero = kx == 0 && jj == 0 && kk == 0;   // zero frequency  F:19924-19926
    const double kpm0 = a.kpm[0][kx], kpm1 = a.kpm[1][jj], kpm2 = a.kpm[2][kk];
    if (NC == 3) {
      const cplx t0 = cscale(a.base.scale, img[idx]), t1 = cscale(a.base.scale, img[comp + idx]), t2 = cscale(a.base.scale, img[2 * comp + idx]);
      cplx e0 = cmake(0.0, 0.0), e1 = e0, e2 = e0;
      if (!zero) g0_point_rcp(t0, t1, t2, kpm0, kpm1, kpm2, a.kp[0][kx], a.kp[1][jj], a.kp[2][kk], a.c10, a.c20, &e0, &e1, &e2);
      img[idx] = e0;
      img[comp + idx] = e1;
      img[2 * comp + idx] = e2;
    } else {
      const double norm_kp2 = kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2;
      img[idx] = zero ? cmake(0.0, 0.0) : cscale(a.base.scale * a.c10 / norm_kp2, img[idx]);
    }
  }
}

// ---- the same phases on the JOINT image [p][NC][C] of the tile's NC components (SmoothPlan::joint): one image of NC * C lines
template <int C, int NC, int B, bool NTL>
FG_HD void smooth_joint_load_impl(const SmoothXArgs& a, int block, int tid, int nthreads, cplx* img) {
  constexpr int W = NC * C;
  const int col0 = block * C, total = a.base.plan.n * W;
  for (int i0 = tid; i0 < total; i0 += B * nthreads) {
    cplx v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int p = idx / W, r = idx % W, c = r / C, t = r % C;
      v[i] = idx < total && col0 + t < a.base.ncols ? smooth_cload<NTL>(&a.base.data[(long)c * a.comp_stride + col0 + (long)p * a.base.ls + t])
                                                    : cmake(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      if (idx < total) img[idx] = v[i];
    }
  }
}

template <int C, int NC, int B>
FG_HD void smooth_joint_load(const SmoothXArgs& a, int block, int tid, int nthreads, cplx* img) {
  if (a.base.nt & 2) smooth_joint_load_impl<C, NC, B, true>(a, block, tid, nthreads, img);
  else smooth_joint_load_impl<C, NC, B, false>(a, block, tid, nthreads, img);
}

template <int C, int NC>
FG_HD void smooth_joint_store(const SmoothXArgs& a, int block, int tid, int nthreads, const cplx* img) {
  constexpr int W = NC * C;
  const int col0 = block * C, total = a.base.plan.n * W;
  for (int idx = tid; idx < total; idx += nthreads) {
    const int p = idx / W, r = idx % W, c = r / C, t = r % C;
    if (col0 + t < a.base.ncols) cstore_stream(&a.base.data[(long)c * a.comp_stride + col0 + (long)p * a.base.ls + t], img[idx], a.base.nt);
  }
}

template <int C, int NC>
FG_HD void smooth_joint_green(const SmoothXArgs& a, int block, int tid, int nthreads, cplx* img) {
  constexpr int W = NC * C;
  const int n = a.base.plan.n, col0 = block * C;
  // a thread's column is the same in every turn of its loop when the stride is a multiple of the tile width: the y and z
  // factors of that column (four of the six table reads of a point, and the division that splits the column index) once
  const bool fixed_col = nthreads % C == 0;
  int jj = 0, kk = 0;
  bool live = false;
  double kpm1 = 0.0, kpm2 = 0.0;
  cplx kp1 = cmake(0.0, 0.0), kp2 = kp1;
  auto column = [&](int t) {
    const int col = col0 + t;
    live = col < a.base.ncols;
    if (!live) return;
    const int jl = smooth_div(col, 1.0 / (double)a.nzc);
    kk = col - jl * a.nzc, jj = a.jj0 + jl;
    live = kk < a.nzf;   // (row padding)
    if (!live) return;
    kpm1 = a.kpm[1][jj], kpm2 = a.kpm[2][kk];
    if (NC == 3) kp1 = a.kp[1][jj], kp2 = a.kp[2][kk];
  };
  if (fixed_col) {
    column(tid % C);
    if (!live) return;
  }
  for (int idx = tid; idx < n * C; idx += nthreads) {
    const int kx = idx / C, t = idx % C;
    if (!fixed_col) {
      column(t);
      if (!live) continue;
    }
    const bool zero = kx == 0 && jj == 0 && kk == 0;   // zero frequency  F:19924-19926
    cplx* q = img + kx * W + t;
    const double kpm0 = a.kpm[0][kx];
    if (NC == 3) {
      const cplx t0 = cscale(a.base.scale, q[0]), t1 = cscale(a.base.scale, q[C]), t2 = cscale(a.base.scale, q[2 * C]);
      cplx e0 = cmake(0.0, 0.0), e1 = e0, e2 = e0;
      if (!zero) g0_point_rcp(t0, t1, t2, kpm0, kpm1, kpm2, a.kp[0][kx], kp1, kp2, a.c10, a.c20, &e0, &e1, &e2);
      q[0] = e0;
      q[C] = e1;
      q[2 * C] = e2;
    } else {   // scalar modes: c10 / |k|^2  (G0OperatorFourierStaggeredHeat F:19758-19823)
      const double norm_kp2 = kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2;
      q[0] = zero ? cmake(0.0, 0.0) : cscale(a.base.scale * a.c10 / norm_kp2, q[0]);
    }
  }
}

// z passes: the tile = rows [row0, row0 + lines), image [l][pitch] (pitch >= M + 1)
// r2c: the packed real row as M complex points -> image
template <int B, bool NTL>
FG_HD void smooth_z_load_packed_impl(const SmoothZArgs& a, long row0, int tid, int nthreads, cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M), total = a.plan.lines * M;
  const double inv = 1.0 / (double)M;
  for (int i0 = tid; i0 < total; i0 += B * nthreads) {
    cplx v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), m = idx - l * M;
      const long row = row0 + l;
      v[i] = idx < total && row < a.nrows ? smooth_cload<NTL>(&reinterpret_cast<const cplx*>(a.data + row * a.nzp)[m]) : cmake(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), m = idx - l * M;
      if (idx < total) img[l * pitch + m] = v[i];
    }
  }
}

template <int B>
FG_HD void smooth_z_load_packed(const SmoothZArgs& a, long row0, int tid, int nthreads, cplx* img) {
  if (a.nt & 2) smooth_z_load_packed_impl<B, true>(a, row0, tid, nthreads, img);
  else smooth_z_load_packed_impl<B, false>(a, row0, tid, nthreads, img);
}

// r2c: the real split X[k], k = 0 .. M, of the transformed image -> memory (FFTW's r2c layout)
template <int B>
FG_HD void smooth_z_split_store(const SmoothZArgs& a, long row0, int tid, int nthreads, const cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M), total = a.plan.lines * (M + 1);
  const double inv = 1.0 / (double)(M + 1);
  for (int i0 = tid; i0 < total; i0 += B * nthreads) {
    cplx wk[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {   // the split roots first: loads in flight together
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * (M + 1);
      wk[i] = idx < total ? a.w[k] : cmake(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * (M + 1);
      const long row = row0 + l;
      if (idx >= total || row >= a.nrows) continue;
      const cplx zk = img[l * pitch + (k == M ? 0 : k)], zmk = img[l * pitch + (k == 0 ? 0 : M - k)];
      cstore_stream(&reinterpret_cast<cplx*>(a.data + row * a.nzp)[k], r2c_split(zk, zmk, wk[i]), a.nt);
    }
  }
}

// c2r: the M + 1 coefficients of a row -> image
template <int B, bool NTL>
FG_HD void smooth_z_load_spectrum_impl(const SmoothZArgs& a, long row0, int tid, int nthreads, cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M), total = a.plan.lines * (M + 1);
  const double inv = 1.0 / (double)(M + 1);
  for (int i0 = tid; i0 < total; i0 += B * nthreads) {
    cplx v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * (M + 1);
      const long row = row0 + l;
      v[i] = idx < total && row < a.nrows ? smooth_cload<NTL>(&reinterpret_cast<const cplx*>(a.data + row * a.nzp)[k]) : cmake(0.0, 0.0);
      if (k == 0 || k == M) v[i].im = 0.0;   // FFTW's c2r ignores the imaginary parts of the DC and Nyquist bins
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * (M + 1);
      if (idx < total) img[l * pitch + k] = v[i];
    }
  }
}

template <int B>
FG_HD void smooth_z_load_spectrum(const SmoothZArgs& a, long row0, int tid, int nthreads, cplx* img) {
  if (a.nt & 2) smooth_z_load_spectrum_impl<B, true>(a, row0, tid, nthreads, img);
  else smooth_z_load_spectrum_impl<B, false>(a, row0, tid, nthreads, img);
}

// c2r: Z'[k] = merge(X[k], X[M - k]), k < M, in place: the merge pairs k and M - k, thread (l, k), k <= M / 2, owns both
template <int B>
FG_HD void smooth_z_merge(const SmoothZArgs& a, int tid, int nthreads, cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M), half = M / 2 + 1, total = a.plan.lines * half;
  const double inv = 1.0 / (double)half;
  for (int i0 = tid; i0 < total; i0 += (B / 2) * nthreads) {
    cplx w1[B / 2], w2[B / 2];
#pragma unroll
    for (int i = 0; i < B / 2; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * half;
      w1[i] = idx < total ? a.w[k] : cmake(0.0, 0.0);
      w2[i] = idx < total ? a.w[M - k] : cmake(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < B / 2; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * half;
      if (idx >= total) continue;
      cplx* row = img + l * pitch;
      const cplx xk = row[k], xm = row[M - k];
      row[k] = c2r_merge(xk, xm, w1[i]);
      if (k != 0 && M - k != k) row[M - k] = c2r_merge(xm, xk, w2[i]);
    }
  }
}

// c2r: the M complex points of the inverse transform = the nz reals of the row -> memory
FG_HD void smooth_z_store_packed(const SmoothZArgs& a, long row0, int tid, int nthreads, const cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M);
  const double inv = 1.0 / (double)M;
  for (int idx = tid; idx < a.plan.lines * M; idx += nthreads) {
    const int l = smooth_div(idx, inv), m = idx - l * M;
    const long row = row0 + l;
    if (row < a.nrows) cstore_stream(&reinterpret_cast<cplx*>(a.data + row * a.nzp)[m], img[l * pitch + m], a.nt);
  }
}

// ---- odd nz: the real row transformed as nz complex points (twice the arithmetic of the packed form, instead of the O(nz^2)
// sums such rows took before), image [l][pitch], pitch >= nz + 1
template <int B>
FG_HD void smooth_zodd_load_real(const SmoothZArgs& a, long row0, int tid, int nthreads, cplx* img) {
  const int nz = a.plan.n, pitch = smooth_z_pitch(nz), total = a.plan.lines * nz;
  const double inv = 1.0 / (double)nz;
  for (int i0 = tid; i0 < total; i0 += B * nthreads) {
    double v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), m = idx - l * nz;
      const long row = row0 + l;
      v[i] = idx < total && row < a.nrows ? a.data[row * a.nzp + m] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), m = idx - l * nz;
      if (idx < total) img[l * pitch + m] = cmake(v[i], 0.0);
    }
  }
}

// r2c: the coefficients k = 0 .. nz / 2 -> memory
FG_HD void smooth_zodd_store_half(const SmoothZArgs& a, long row0, int tid, int nthreads, const cplx* img) {
  const int nz = a.plan.n, pitch = smooth_z_pitch(nz), nzf = nz / 2 + 1;
  const double inv = 1.0 / (double)nzf;
  for (int idx = tid; idx < a.plan.lines * nzf; idx += nthreads) {
    const int l = smooth_div(idx, inv), k = idx - l * nzf;
    const long row = row0 + l;
    if (row < a.nrows) cstore_stream(&reinterpret_cast<cplx*>(a.data + row * a.nzp)[k], img[l * pitch + k], a.nt);
  }
}

// c2r: the coefficients k = 0 .. nz / 2 and their mirror images conj X[k] at nz - k -> image
template <int B, bool NTL>
FG_HD void smooth_zodd_load_half_impl(const SmoothZArgs& a, long row0, int tid, int nthreads, cplx* img) {
  const int nz = a.plan.n, pitch = smooth_z_pitch(nz), nzf = nz / 2 + 1, total = a.plan.lines * nzf;
  const double inv = 1.0 / (double)nzf;
  for (int i0 = tid; i0 < total; i0 += B * nthreads) {
    cplx v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * nzf;
      const long row = row0 + l;
      v[i] = idx < total && row < a.nrows ? smooth_cload<NTL>(&reinterpret_cast<const cplx*>(a.data + row * a.nzp)[k]) : cmake(0.0, 0.0);
      if (k == 0) v[i].im = 0.0;   // FFTW's c2r ignores the imaginary part of the DC bin
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int idx = i0 + i * nthreads;
      const int l = smooth_div(idx, inv), k = idx - l * nzf;
      if (idx >= total) continue;
      img[l * pitch + k] = v[i];
      if (k > 0) img[l * pitch + nz - k] = cconj(v[i]);
    }
  }
}

template <int B>
FG_HD void smooth_zodd_load_half(const SmoothZArgs& a, long row0, int tid, int nthreads, cplx* img) {
  if (a.nt & 2) smooth_zodd_load_half_impl<B, true>(a, row0, tid, nthreads, img);
  else smooth_zodd_load_half_impl<B, false>(a, row0, tid, nthreads, img);
}

// c2r: the real parts of the inverse transform -> memory
FG_HD void smooth_zodd_store_real(const SmoothZArgs& a, long row0, int tid, int nthreads, const cplx* img) {
  const int nz = a.plan.n, pitch = smooth_z_pitch(nz);
  const double inv = 1.0 / (double)nz;
  for (int idx = tid; idx < a.plan.lines * nz; idx += nthreads) {
    const int l = smooth_div(idx, inv), m = idx - l * nz;
    const long row = row0 + l;
    if (row < a.nrows) a.data[row * a.nzp + m] = img[l * pitch + m].re;
  }
}

}  // namespace fft
}  // namespace fg
