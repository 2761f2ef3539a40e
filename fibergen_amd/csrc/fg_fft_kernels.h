// Per-thread bodies of the FFT kernels, written as barrier-separated phases
// (see fg_fft_core.h) so that fg_fft.hip (device) and tests/emulate (host)
// run identical code.
#pragma once

#include "fg_fft_core.h"

namespace fg {
namespace fft {

// Launch geometry shared by device and emulation.
// Columns per tile of a strided pass: 8 (128-B segments), more for short lines
// so that a block has >= 256 threads.
template <int N>
struct TileCols {
  static constexpr int value = (2048 / N) > 8 ? (2048 / N) : 8;
};
// Rows per block in the z pass (M = nz/2 packed points per row): aim at 256 threads.
template <int M>
struct ZLines {
  static constexpr int value = (2048 / M) > 4 ? (2048 / M) : 4;
};

// ------------------------------------------------------------------ strided c2c
// Lines of N points, `ls` complex apart, for `ncols` adjacent columns (stride 1)
// and `nouter` repetitions `os` apart.  A block transforms a tile of C columns.
//   y pass: ls = nzc,     ncols = nzc,     os = ny*nzc, nouter = nx
//   x pass: ls = ny*nzc,  ncols = ny*nzc,  os = 0,      nouter = 1
struct StridedArgs {
  cplx* data;
  long ls, os;
  int ncols, tiles_per_outer;
  double scale;     // applied at the store (1.0 = none)
  const cplx* tw;   // pass twiddles of N
};

template <int N, int C, int DIR>
struct StridedKernel {
  static constexpr int T = N / 8;
  static constexpr int THREADS = T * C;
  static constexpr int PN = N + N / 8;
  static constexpr int LDS_DOUBLES = 2 * PN * C;
  static constexpr int NPHASE = Line<N>::NPHASE;
  struct Regs {
    cplx v[8];
    long base;
    int jt, t;
    bool valid;
  };
  template <int PH>
  FG_HD static void phase(Regs& r, int block, int tid, double* lds, const StridedArgs& a) {
    const LdsMap L = {C, 1, PN * C};
    if (PH == 0) {
      r.t = tid % C;
      r.jt = tid / C;
      int o = block / a.tiles_per_outer;
      int col = (block % a.tiles_per_outer) * C + r.t;
      r.valid = col < a.ncols;
      r.base = (long)o * a.os + col;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        r.v[q] = r.valid ? a.data[r.base + (long)Line<N>::first_index(r.jt, q) * a.ls] : cmake(0.0, 0.0);
    }
    Line<N>::template phase<DIR, PH>(r.v, r.jt, lds, L, r.t, a.tw);
    if (PH == NPHASE - 1 && r.valid) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        cplx o = r.v[q];
        if (a.scale != 1.0) o = cscale(a.scale, o);
        a.data[r.base + (long)Line<N>::last_index(r.jt, q) * a.ls] = o;
      }
    }
  }
};

// ------------------------------------------------------------------ z pass, r2c
// Rows of nz = 2M reals (row stride nzp doubles) -> nzc = M+1 complex, in place.
struct ZArgs {
  double* data;     // component base
  long nrows;       // nx*ny
  int nzp;
  const cplx* tw;   // pass twiddles of M
  const cplx* wz;   // e^{-2 pi i k/nz}, k = 0..M
};

template <int M, int LINES>
struct R2CKernel {
  static constexpr int T = M / 8;
  static constexpr int THREADS = T * LINES;
  static constexpr int LS = M + M / 8 + 2;       // line stride (doubles)
  static constexpr int LDS_DOUBLES = 2 * LS * LINES;
  static constexpr int NPHASE = Line<M>::NPHASE + 1;
  struct Regs {
    cplx v[8];
    double* row;
    int jt, l;
    bool valid;
  };
  template <int PH>
  FG_HD static void phase(Regs& r, int block, int tid, double* lds, const ZArgs& a) {
    const LdsMap L = {1, LS, LS * LINES};
    if (PH == 0) {
      r.jt = tid % T;
      r.l = tid / T;
      long row = (long)block * LINES + r.l;
      r.valid = row < a.nrows;
      r.row = a.data + row * a.nzp;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int m = Line<M>::first_index(r.jt, q);
        r.v[q] = r.valid ? reinterpret_cast<const cplx*>(r.row)[m] : cmake(0.0, 0.0);
      }
    }
    if (PH < NPHASE - 1) {
      Line<M>::template phase<-1, (PH < NPHASE - 1 ? PH : 0)>(r.v, r.jt, lds, L, r.l, a.tw);
      if (PH == NPHASE - 2) {  // natural-order spectrum of the packed line -> LDS
#pragma unroll
        for (int q = 0; q < 8; ++q) lds_put(lds, L, Line<M>::last_index(r.jt, q), r.l, r.v[q]);
      }
    } else if (r.valid) {
      cplx* out = reinterpret_cast<cplx*>(r.row);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int k = r.jt + q * T;
        cplx zk = lds_get(lds, L, k, r.l);
        cplx zm = lds_get(lds, L, (M - k) % M, r.l);
        out[k] = r2c_split(zk, zm, a.wz[k]);
      }
      if (r.jt == 0) {  // k = M (Nyquist): Z[M] := Z[0]
        cplx z0 = lds_get(lds, L, 0, r.l);
        out[M] = r2c_split(z0, z0, a.wz[M]);
      }
    }
  }
};

// ------------------------------------------------------------------ z pass, c2r
template <int M, int LINES>
struct C2RKernel {
  static constexpr int T = M / 8;
  static constexpr int THREADS = T * LINES;
  static constexpr int LS = M + M / 8 + 2;
  static constexpr int LDS_DOUBLES = 2 * LS * LINES;
  static constexpr int NPHASE = Line<M>::NPHASE;
  struct Regs {
    cplx v[8];
    double* row;
    int jt, l;
    bool valid;
  };
  template <int PH>
  FG_HD static void phase(Regs& r, int block, int tid, double* lds, const ZArgs& a) {
    const LdsMap L = {1, LS, LS * LINES};
    if (PH == 0) {
      r.jt = tid % T;
      r.l = tid / T;
      long row = (long)block * LINES + r.l;
      r.valid = row < a.nrows;
      r.row = a.data + row * a.nzp;
      const cplx* in = reinterpret_cast<const cplx*>(r.row);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int m = Line<M>::first_index(r.jt, q);
        cplx xk = cmake(0.0, 0.0), xm = cmake(0.0, 0.0);
        if (r.valid) {
          xk = in[m];
          xm = in[M - m];
          // FFTW's c2r ignores the imaginary parts of the DC and Nyquist bins
          if (m == 0) { xk.im = 0.0; xm.im = 0.0; }
        }
        r.v[q] = c2r_merge(xk, xm, a.wz[m]);
      }
    }
    Line<M>::template phase<+1, PH>(r.v, r.jt, lds, L, r.l, a.tw);
    if (PH == NPHASE - 1 && r.valid) {
      cplx* out = reinterpret_cast<cplx*>(r.row);
#pragma unroll
      for (int q = 0; q < 8; ++q) out[Line<M>::last_index(r.jt, q)] = r.v[q];
    }
  }
};

// ------------------------------------------------------------------ generic O(n^2) fall-backs
// Any length (odd, prime, tiny): one thread per output point, out of place.
// w[k] = e^{-2 pi i k/n}.  Used for axes that are not a power of two in [8,1024].
FG_HD cplx wpow(const cplx* w, long n, long e, int dir) {
  cplx c = w[e % n];
  return dir > 0 ? cconj(c) : c;
}

// c2c along a strided axis: dst[base + k*ls] = scale * sum_n src[base + n*ls] w^{dir nk}
FG_HD void dft_strided_point(const cplx* src, cplx* dst, long base, long ls, int n, int k, int dir,
                             double scale, const cplx* w) {
  double sr = 0.0, si = 0.0;
  for (int m = 0; m < n; ++m) {
    cplx x = src[base + (long)m * ls];
    cplx c = wpow(w, n, (long)m * k, dir);
    sr += x.re * c.re - x.im * c.im;
    si += x.re * c.im + x.im * c.re;
  }
  dst[base + (long)k * ls] = cmake(scale * sr, scale * si);
}

// r2c along z: out[k] = sum_n x[n] w^{nk}, k = 0..nz/2
FG_HD void r2c_point(const double* xrow, cplx* orow, int nz, int k, const cplx* w) {
  double sr = 0.0, si = 0.0;
  for (int m = 0; m < nz; ++m) {
    cplx c = wpow(w, nz, (long)m * k, -1);
    sr += xrow[m] * c.re;
    si += xrow[m] * c.im;
  }
  orow[k] = cmake(sr, si);
}

// c2r along z (unnormalised): x[n] = Re X0 + 2 sum_{0<k<nz/2} Re(X_k e^{+2 pi i nk/nz}) [+ (-1)^n Re X_{nz/2}]
FG_HD void c2r_point(const cplx* xrow, double* orow, int nz, int m, const cplx* w) {
  double s = xrow[0].re;
  int kmax = (nz - 1) / 2;
  for (int k = 1; k <= kmax; ++k) {
    cplx c = wpow(w, nz, (long)m * k, +1);
    s += 2.0 * (xrow[k].re * c.re - xrow[k].im * c.im);
  }
  if (nz % 2 == 0 && nz > 1) s += (m % 2 == 0 ? 1.0 : -1.0) * xrow[nz / 2].re;
  orow[m] = s;
}

}  // namespace fft
}  // namespace fg
