// Transform passes for lengths that are NOT a power of two (nor 3, 5, 7, 9 times one): N = a product of the radices
// 8, 4, 2, 3, 5, 7, 11, 13 -- the decimal grid sizes users of the reference pick (100, 200, 300, 400, 500 ...; FFTW takes any).
//
// Stockham autosort on a tile of lines held in LDS: a workgroup loads C whole lines (C adjacent columns for the strided
// y / x passes: one 128-byte segment per line point for C = 8; C rows for the z passes), runs one pass per radix between two
// LDS images (pass with radix R, Ns = product of the radices before it:  butterfly j < N/R reads points j + r N/R, multiplies
// by w^{r (j mod Ns) N/(Ns R)}, transforms, writes points (j / Ns) Ns R + (j mod Ns) + r Ns -- natural order after the last
// pass), and writes the tile back: one read and one write of the data per axis, like the power-of-two kernels (which stay
// the faster ones for their lengths: 8 points per thread in registers, fewer LDS sweeps).
// The per-thread code is FG_HD: tests/emulate runs it on the host against numpy.
#pragma once

#include "fg_fft_core.h"

namespace fg {
namespace fft {

constexpr int kSmoothMaxFactors = 12;

struct SmoothPlan {
  int n = 0;       // line length (z passes: nz / 2)
  int nfac = 0;
  int fac[kSmoothMaxFactors] = {};
};

// radices in pass order; false when n has a prime factor above 13 (or too many factors)
inline bool smooth_plan(int n, SmoothPlan* p) {
  p->n = n;
  p->nfac = 0;
  if (n < 2) return false;
  int m = n;
  auto take = [&](int r) {
    while (m % r == 0) {
      if (p->nfac >= kSmoothMaxFactors) return false;
      p->fac[p->nfac++] = r;
      m /= r;
    }
    return true;
  };
  // odd radices first (their twiddle-free first pass is the expensive butterfly), then 8s, one 4 or 2
  for (int r : {13, 11, 7, 5, 3})
    if (!take(r)) return false;
  while (m % 8 == 0) {
    if (p->nfac >= kSmoothMaxFactors) return false;
    p->fac[p->nfac++] = 8;
    m /= 8;
  }
  if (!take(4) || !take(2)) return false;
  return m == 1;
}

// cos / sin (2 pi j / R), j = 1 .. (R - 1) / 2 (local constant tables: usable from host and device code alike)
template <int R> FG_HD double odd_cos(int j);
template <int R> FG_HD double odd_sin(int j);
template <> FG_HD double odd_cos<3>(int j) {
  constexpr double c[1] = {-0.5};
  return c[j - 1];
}
template <> FG_HD double odd_sin<3>(int j) {
  constexpr double s[1] = {0.8660254037844386467637232};
  return s[j - 1];
}
template <> FG_HD double odd_cos<5>(int j) {
  constexpr double c[2] = {0.3090169943749474241022934, -0.8090169943749474241022934};
  return c[j - 1];
}
template <> FG_HD double odd_sin<5>(int j) {
  constexpr double s[2] = {0.9510565162951535721164393, 0.587785252292473129168706};
  return s[j - 1];
}
template <> FG_HD double odd_cos<7>(int j) {
  constexpr double c[3] = {0.6234898018587335305250049, -0.2225209339563144042889026, -0.9009688679024191262361023};
  return c[j - 1];
}
template <> FG_HD double odd_sin<7>(int j) {
  constexpr double s[3] = {0.7818314824680298087084445, 0.9749279121818236070181317, 0.4338837391175581204757683};
  return s[j - 1];
}
template <> FG_HD double odd_cos<11>(int j) {
  constexpr double c[5] = {0.8412535328311811688618116, 0.4154150130018864255292741, -0.1423148382732851404437927, -0.6548607339452850640569251, -0.9594929736144973898903681};
  return c[j - 1];
}
template <> FG_HD double odd_sin<11>(int j) {
  constexpr double s[5] = {0.540640817455597582107636, 0.9096319953545183714117154, 0.989821441880932732376092, 0.7557495743542582837740358, 0.2817325568414296977114179};
  return s[j - 1];
}
template <> FG_HD double odd_cos<13>(int j) {
  constexpr double c[6] = {0.8854560256532098959003755, 0.5680647467311558025118076, 0.1205366802553230533490677, -0.3546048870425356259696379, -0.7485107481711010986346306, -0.9709418174260520271569823};
  return c[j - 1];
}
template <> FG_HD double odd_sin<13>(int j) {
  constexpr double s[6] = {0.4647231720437685456560153, 0.8229838658936563945796174, 0.9927088740980539928007516, 0.9350162426854148234397846, 0.6631226582407952023767855, 0.2393156642875577671487537};
  return s[j - 1];
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
      const int j = (m * k) % R;                       // compile-time after unrolling
      const double cj = odd_cos<R>(j <= H ? j : R - j);
      const double sj = (j <= H ? 1.0 : -1.0) * odd_sin<R>(j <= H ? j : R - j);
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

template <int R, int DIR>
FG_HD void dft_any(cplx* v) {
  if constexpr (R == 2) dft2<DIR>(v);
  else if constexpr (R == 4) dft4<DIR>(v);
  else if constexpr (R == 8) dft8<DIR>(v);
  else dft_odd<R, DIR>(v);
}

// Addressing of a tile in LDS: point p of line t sits at p * sp + t * sc.
//   strided passes: sp = C, sc = 1 (the C columns of a line point are adjacent, as in memory); threads run over t fastest
//   z passes:       sp = 1, sc = line pitch (a row is contiguous, as in memory);                threads run over j fastest
struct SmoothMap {
  int sp, sc, lines;
  bool jfast;
};

// One pass for the thread `tid` of `nthreads`.  w: e^{-2 pi i k / (N * wscale)}, entry k * wscale = the N-th root's power k.
template <int R, int DIR>
FG_HD void smooth_pass(const cplx* in, cplx* out, int N, int Ns, const SmoothMap& L, const cplx* w, int wscale, int tid, int nthreads) {
  const int nb = N / R;
  const int tws = (N / (Ns * R)) * wscale;
  for (int idx = tid; idx < nb * L.lines; idx += nthreads) {
    const int t = L.jfast ? idx / nb : idx % L.lines;
    const int j = L.jfast ? idx % nb : idx / L.lines;
    const int k = j % Ns;
    const cplx* src = in + (long)t * L.sc;
    cplx v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[(j + r * nb) * L.sp];
    if (Ns > 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) {
        const cplx tw = w[(long)r * k * tws];
        v[r] = cmul(v[r], DIR < 0 ? tw : cconj(tw));
      }
    }
    dft_any<R, DIR>(v);
    cplx* dst = out + (long)t * L.sc + (long)((j / Ns) * Ns * R + k) * L.sp;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[(long)r * Ns * L.sp] = v[r];
  }
}

template <int DIR>
FG_HD void smooth_pass_any(int R, const cplx* in, cplx* out, int N, int Ns, const SmoothMap& L, const cplx* w, int wscale, int tid,
                           int nthreads) {
  switch (R) {
    case 2: smooth_pass<2, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
    case 3: smooth_pass<3, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
    case 4: smooth_pass<4, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
    case 5: smooth_pass<5, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
    case 7: smooth_pass<7, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
    case 8: smooth_pass<8, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
    case 11: smooth_pass<11, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
    default: smooth_pass<13, DIR>(in, out, N, Ns, L, w, wscale, tid, nthreads); break;
  }
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

struct SmoothZArgs {
  double* data;        // component base (padded real rows / complex rows)
  long nrows;
  int nzp;
  const cplx* w;       // e^{-2 pi i k / nz}, k < nz  (its even entries are the roots of nz / 2)
  int nt;
  SmoothPlan plan;     // of M = nz / 2
};

// ---- tile phases (between two workgroup barriers each; `tid` of `nthreads`)
// strided pass: the tile = columns [col0, col0 + C) of outer index o, image [p][C]
template <int C>
FG_HD void smooth_strided_load(const SmoothArgs& a, int block, int tid, int nthreads, cplx* img) {
  const int o = block / a.tiles_per_outer, col0 = (block % a.tiles_per_outer) * C;
  const long base = (long)o * a.os + col0;
  for (int idx = tid; idx < a.plan.n * C; idx += nthreads) {
    const int p = idx / C, t = idx % C;
    img[idx] = col0 + t < a.ncols ? cload_stream(&a.data[base + (long)p * a.ls + t], a.nt) : cmake(0.0, 0.0);
  }
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

// z passes: the tile = rows [row0, row0 + lines), image [l][pitch] (pitch >= M + 1)
FG_HD int smooth_z_pitch(int M) { return M + 1 + ((M + 1) % 2 == 0 ? 1 : 0); }   // odd: rows start on different banks

// r2c: the packed real row as M complex points -> image
FG_HD void smooth_z_load_packed(const SmoothZArgs& a, long row0, int lines, int tid, int nthreads, cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M);
  for (int idx = tid; idx < lines * M; idx += nthreads) {
    const int l = idx / M, m = idx % M;
    const long row = row0 + l;
    img[l * pitch + m] = row < a.nrows ? cload_stream(&reinterpret_cast<const cplx*>(a.data + row * a.nzp)[m], a.nt) : cmake(0.0, 0.0);
  }
}

// r2c: the real split X[k], k = 0 .. M, of the transformed image -> memory (FFTW's r2c layout)
FG_HD void smooth_z_split_store(const SmoothZArgs& a, long row0, int lines, int tid, int nthreads, const cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M);
  for (int idx = tid; idx < lines * (M + 1); idx += nthreads) {
    const int l = idx / (M + 1), k = idx % (M + 1);
    const long row = row0 + l;
    if (row >= a.nrows) continue;
    const cplx zk = img[l * pitch + (k == M ? 0 : k)], zmk = img[l * pitch + (k == 0 ? 0 : M - k)];
    cstore_stream(&reinterpret_cast<cplx*>(a.data + row * a.nzp)[k], r2c_split(zk, zmk, a.w[k]), a.nt);
  }
}

// c2r: the M + 1 coefficients of a row -> image
FG_HD void smooth_z_load_spectrum(const SmoothZArgs& a, long row0, int lines, int tid, int nthreads, cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M);
  for (int idx = tid; idx < lines * (M + 1); idx += nthreads) {
    const int l = idx / (M + 1), k = idx % (M + 1);
    const long row = row0 + l;
    cplx x = row < a.nrows ? cload_stream(&reinterpret_cast<const cplx*>(a.data + row * a.nzp)[k], a.nt) : cmake(0.0, 0.0);
    if (k == 0 || k == M) x.im = 0.0;   // FFTW's c2r ignores the imaginary parts of the DC and Nyquist bins
    img[l * pitch + k] = x;
  }
}

// c2r: Z'[k] = merge(X[k], X[M - k]), k < M, from one image into the other
FG_HD void smooth_z_merge(const SmoothZArgs& a, int lines, int tid, int nthreads, const cplx* in, cplx* out) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M);
  for (int idx = tid; idx < lines * M; idx += nthreads) {
    const int l = idx / M, k = idx % M;
    out[l * pitch + k] = c2r_merge(in[l * pitch + k], in[l * pitch + M - k], a.w[k]);
  }
}

// c2r: the M complex points of the inverse transform = the nz reals of the row -> memory
FG_HD void smooth_z_store_packed(const SmoothZArgs& a, long row0, int lines, int tid, int nthreads, const cplx* img) {
  const int M = a.plan.n, pitch = smooth_z_pitch(M);
  for (int idx = tid; idx < lines * M; idx += nthreads) {
    const int l = idx / M, m = idx % M;
    const long row = row0 + l;
    if (row < a.nrows) cstore_stream(&reinterpret_cast<cplx*>(a.data + row * a.nzp)[m], img[l * pitch + m], a.nt);
  }
}

}  // namespace fft
}  // namespace fg
