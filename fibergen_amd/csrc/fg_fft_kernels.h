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

// Tile width of the fused x pass: 256-thread blocks, so that two blocks share a CU at 256 VGPRs
// and one block's loads / barriers overlap the other's butterflies.
template <int N>
struct XTileCols {
  static constexpr int value = (2048 / N) > 8 ? (2048 / N) : 8;  // 64-B segments (C = 4) measured 35 % slower
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
  int nt;           // streaming (cache-bypassing) stores, see cstore_stream
  int xcd_order;    // 1: blockIdx is remapped so that each XCD works on a contiguous run of tiles
  // Slab decomposition: the y pass writes (forward) / reads (inverse) the all-to-all layout
  // [peer q][x][ky mod nyl][kz] directly, so the pencil transpose needs no pack / unpack sweeps.  Line point j of the
  // re-mapped side sits at  j*ls + (j >> split) * jump  (split = log2 nyl, jump = block stride - nyl*ls); the other
  // side keeps the plain layout.  Defaults = in place, plain layout on both sides.
  cplx* out = nullptr;       // nullptr: in place
  long out_cs = 0;           // complex elements between components of `out` (out != nullptr)
  long os_out = 0;           // outer stride on the store side (out != nullptr)
  int split_in = 31, split_out = 31;
  long jump_in = 0, jump_out = 0;
  // x-contiguous intermediate layout (Fft3::c2c_y_xlayout): one side of the y pass addresses [zc/8][y][x][8] instead of
  // [x][y][zc] -- its line stride differs (ls_out; 0 = ls) and tile z of outer index o starts at o*os + z*ts (ts = 0: the
  // plain layout's o*os + z*C)
  long ls_out = 0;
  long ts_in = 0, ts_out = 0;
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
    long base, obase;
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
      const int tile = block % a.tiles_per_outer;
      int col = tile * C + r.t;
      r.valid = col < a.ncols;
      r.base = (long)o * a.os + (a.ts_in ? (long)tile * a.ts_in + r.t : (long)col);
      r.obase = a.out ? (long)o * a.os_out + (a.ts_out ? (long)tile * a.ts_out + r.t : (long)col) : r.base;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = Line<N>::first_index(r.jt, q);
        r.v[q] = r.valid ? cload_stream(&a.data[r.base + (long)j * a.ls + (long)(j >> a.split_in) * a.jump_in], a.nt)
                         : cmake(0.0, 0.0);
      }
    }
    Line<N>::template phase<DIR, PH>(r.v, r.jt, lds, L, r.t, a.tw);
    if (PH == NPHASE - 1 && r.valid) {
      cplx* const dst = a.out ? a.out : a.data;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        cplx o = r.v[q];
        if (a.scale != 1.0) o = cscale(a.scale, o);
        const int j = Line<N>::last_index(r.jt, q);
        cstore_stream(&dst[r.obase + (long)j * (a.ls_out ? a.ls_out : a.ls) + (long)(j >> a.split_out) * a.jump_out], o, a.nt);
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
  int nt;           // streaming stores
};

// MIRROR: the split X[k] from Z[k] and Z[M - k] takes the mirrored value from the lane that holds it (wave shuffle inside
// the T lanes of the line) right after the last pass, instead of a round trip of the whole spectrum through LDS.
template <int M, int LINES, bool MIRROR = false>
struct R2CKernel {
  static constexpr int T = M / 8;
  static constexpr int THREADS = T * LINES;
  static constexpr int LS = M + M / 8 + 2;       // line stride (doubles)
  static constexpr int LDS_DOUBLES = 2 * LS * LINES;
  static constexpr bool SHUFFLE = MIRROR && T <= 64 && T >= 2;
  static constexpr int NPHASE = Line<M>::NPHASE + (SHUFFLE ? 0 : 1);
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
        r.v[q] = r.valid ? cload_stream(&reinterpret_cast<const cplx*>(r.row)[m], a.nt) : cmake(0.0, 0.0);
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SHUFFLE) {
      Line<M>::template phase<-1, PH>(r.v, r.jt, lds, L, r.l, a.tw);
      if (PH == NPHASE - 1) {
        // slot q = (b, rr) of the last pass (radix RL) holds Z[jt + s T], s = b + rr (8 / RL); Z[M - k] is slot s' = 7 - s of
        // lane T - jt of this line (jt = 0: the own slot 8 - s, Z[M] := Z[0])
        constexpr int RL = pass_radix(M, num_passes(M) - 1), G = 8 / RL;
        const int lane = threadIdx.x & 63;
        const int src = (lane & ~(T - 1)) | ((T - r.jt) & (T - 1));
        cplx* out = reinterpret_cast<cplx*>(r.row);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          constexpr int dummy = 0;
          (void)dummy;
          const int s = q / RL + (q % RL) * G;
          const int sm = 7 - s, qm = (sm % G) * RL + sm / G;
          const int s0 = (8 - s) % 8, q0 = (s0 % G) * RL + s0 / G;
          cplx zm = cmake(__shfl(r.v[qm].re, src), __shfl(r.v[qm].im, src));
          if (r.jt == 0) zm = r.v[q0];
          const int k = r.jt + s * T;
          if (r.valid) cstore_stream(&out[k], r2c_split(r.v[q], zm, a.wz[k]), a.nt);
        }
        if (r.valid && r.jt == 0) cstore_stream(&out[M], r2c_split(r.v[0], r.v[0], a.wz[M]), a.nt);   // k = M (Nyquist): Z[M] := Z[0]
      }
      return;
    }
#endif
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
        cstore_stream(&out[k], r2c_split(zk, zm, a.wz[k]), a.nt);
      }
      if (r.jt == 0) {  // k = M (Nyquist): Z[M] := Z[0]
        cplx z0 = lds_get(lds, L, 0, r.l);
        cstore_stream(&out[M], r2c_split(z0, z0, a.wz[M]), a.nt);
      }
    }
  }
};

// ------------------------------------------------------------------ z pass, c2r
// MIRROR: the mirrored coefficient X[M - m] of the merge comes from the lane that loaded it as ITS X[m] (wave shuffle
// inside the T lanes of the line) instead of a second global load: every coefficient is read once.
template <int M, int LINES, bool MIRROR = false>
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
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (MIRROR && T <= 64 && T >= 2 && pass_radix(M, 0) == 8) {
        // first_index(jt, q) = jt + q T, so X[M - m] = X[(T - jt) + (7 - q) T]: slot 7 - q of lane T - jt of this line
        // (jt = 0: the own slots 8 - q, and the Nyquist bin for q = 0)
        cplx x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = r.valid ? cload_stream(&in[r.jt + q * T], a.nt) : cmake(0.0, 0.0);
        cplx ny = (r.valid && r.jt == 0) ? cload_stream(&in[M], a.nt) : cmake(0.0, 0.0);
        const int lane = threadIdx.x & 63;
        const int src = (lane & ~(T - 1)) | ((T - r.jt) & (T - 1));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          cplx xm = cmake(__shfl(x[7 - q].re, src), __shfl(x[7 - q].im, src));
          if (r.jt == 0) xm = q == 0 ? ny : x[8 - (q ? q : 8)];
          cplx xk = x[q];
          if (r.jt == 0 && q == 0) { xk.im = 0.0; xm.im = 0.0; }   // FFTW's c2r ignores the imaginary parts of DC and Nyquist
          r.v[q] = c2r_merge(xk, xm, a.wz[r.jt + q * T]);
        }
      } else
#endif
      {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int m = Line<M>::first_index(r.jt, q);
        cplx xk = cmake(0.0, 0.0), xm = cmake(0.0, 0.0);
        if (r.valid) {
          xk = cload_stream(&in[m], a.nt);
          xm = cload_stream(&in[M - m], a.nt);
          // FFTW's c2r ignores the imaginary parts of the DC and Nyquist bins
          if (m == 0) { xk.im = 0.0; xm.im = 0.0; }
        }
        r.v[q] = c2r_merge(xk, xm, a.wz[m]);
      }
      }
    }
    Line<M>::template phase<+1, PH>(r.v, r.jt, lds, L, r.l, a.tw);
    if (PH == NPHASE - 1 && r.valid) {
      cplx* out = reinterpret_cast<cplx*>(r.row);
#pragma unroll
      for (int q = 0; q < 8; ++q) cstore_stream(&out[Line<M>::last_index(r.jt, q)], r.v[q], a.nt);
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

// ------------------------------------------------------------------ fused x pass
// forward x-FFT (x 1/N)  ->  Green operator G0  ->  inverse x-FFT, for the 3 components of a
// tile of C columns, in one kernel: the spectrum never leaves the registers, which removes the
// Green-operator sweep and one write + one read of the three components per iteration
// (SURVEY 8d "maximum legal fusion").  G0OperatorFourierStaggeredGeneral  F:19834-19927.
#include "fg_stage_math.h"

namespace fg {
namespace fft {

struct XFusedArgs {
  cplx* data;          // component 0
  long comp_stride;    // complex elements between components
  long ls, os;         // line stride / outer stride (complex elements)
  int ncols, tiles_per_outer;
  int flat_cols;       // 1: columns are the flattened (ky,kz) index of [nx][ny][nzc]; 0: column = kz, outer = ky - jj0
  int nzc, nzf, jj0;
  double scale;        // 1/N applied to the forward spectrum (F:18501-18506)
  double c10, c20;
  const cplx* tw;
  const double* kpm[3];
  const cplx* kp[3];
  int nt;              // streaming stores
  // The Green-operator factors of the transformed axis are rebuilt in the kernel instead of being looked up per point
  // (8 dependent table round trips per thread were 12-16 % of a tile): with theta = pi kx / N,
  //   kpm0^2 = (sin theta / h)^2,  kp0 = (sin theta / h) e^{i theta}   (F:19838-19876; the reference's signed index
  //   shifts theta by -pi for kx > N/2, which changes neither quantity),
  // and kx = jt + off(q):  e^{i theta} = half_root[jt] * xq[q].
  const cplx* half_root;   // e^{-i pi j / N}, j < N/8 (conjugated in the kernel)
  cplx xq[8];              // e^{+i pi off(q) / N}, off(q) = last_index(0, q)
  double inv_h;            // 1/h = 2 N / d of the transformed axis
  int xcd_order;           // 1: blockIdx is remapped so that each XCD works on a contiguous run of tiles
  // Slab decomposition with the three components of a peer's block interleaved (y-slab [p][c][nx/P][ny/P][nzc], what one
  // message per peer delivers): line point j sits at  j*ls + (j >> xsplit) * xjump  (XFusedKernel<.., XSPLIT = true> only)
  int xsplit = 31;
  long xjump = 0;
  // x-contiguous layout [zc/8][y][x][8] (xl_ny = ny > 0; C = 8): tile b = zt * ny + y is the contiguous run of N * 8
  // complex values at b * N * 8, line point j at + 8 j
  int xl_ny = 0;
};

// NC = 3: the three components of the elastic problem and G0OperatorFourierStaggeredGeneral; NC = 1: the scalar modes
// (one potential, G0OperatorFourierStaggeredGeneralHeat  F:19779-19823: c1 = c10 / |k|^2).
// NTC >= 0: the streaming flags are the compile-time constant NTC and the loads of a tile carry no branch (columns past the
// end read the tile's first column and are never stored), so that the first transform waits for ITS component's loads only
// (s_waitcnt vmcnt(16)) instead of for all of them; NTC = -1: flags from XFusedArgs::nt at run time.
template <int N, int C, int NC = 3, bool XSPLIT = false, int NTC = -1>
struct XFusedKernel {
  static constexpr int T = N / 8;
  static constexpr int THREADS = T * C;
  static constexpr int PN = N + N / 8;
  static constexpr int NPL = Line<N>::NPHASE;   // phases of one line transform
  // with two exchanges per transform the LDS image is double-buffered: the barrier after a gather
  // can then be dropped (the next scatter goes to the other buffer), 12 instead of 24 per tile
  // (one component, lines up to 256: ONE exchange image -- its few registers leave room for a third workgroup per CU, which the
  // second image's 37 KB of LDS would take away: r5, old / new library alternating in one job, porous mode: 128^3 K4 15.7 -> 13.9 us,
  // 14 296 -> 14 547 it/s; 256^3 0.0749 -> 0.0724 ms, 2 818 -> 2 838 it/s; 512^3 0.685 -> 0.689 ms: keeps two images)
  static constexpr bool PINGPONG = num_passes(N) == 3 && (NC == 3 || N >= 512);
  static constexpr int BUF_DOUBLES = 2 * PN * C;
  // The pass twiddles sit in LDS behind the exchange buffers (N = 512: 14 KB, the tile then uses 158 of 160 KB).
  // A twiddle read through the vector memory path shares its in-order counter (vmcnt) with the tile's loads and
  // stores, so every twiddled pass waited for the stores of the previous component to be acknowledged
  // (N = 512: 2.55 -> 1.85 ms; for the same reason a workgroup handles ONE tile: in a loop over tiles the next
  // tile's loads would queue behind the stores of the previous one).
  static constexpr bool TW_LDS = N >= 64;   // (r5: for one component too)
  static constexpr int TW_OFF = (PINGPONG ? 2 : 1) * BUF_DOUBLES;
  static constexpr int LDS_DOUBLES = TW_OFF + (TW_LDS ? 2 * tw_total(N) : 0);
  static constexpr int NPHASE = 2 * NC * NPL;   // NC forward + NC inverse transforms
  struct Regs {
    cplx v[NC][8];
    long base;
    int jt, t, jj, kk;
    bool valid;
  };
  FG_HD static void locate(int block, int t, const XFusedArgs& a, long* base, int* jj, int* kk, bool* valid) {
    if (a.xl_ny) {
      const int zt = block / a.xl_ny;
      *jj = a.jj0 + (block - zt * a.xl_ny);
      *kk = zt * C + t;
      *base = (long)block * N * C + t;
      *valid = *kk < a.nzc;
      return;
    }
    const int o = block / a.tiles_per_outer;
    const int col = (block % a.tiles_per_outer) * C + t;
    *valid = col < a.ncols;
    *base = (long)o * a.os + col;
    if (a.flat_cols) {
      const int jl = col / a.nzc;
      *jj = a.jj0 + jl;           // jj0 != 0: y-slab [nx][ny/P][nzc] of a slab-decomposed grid
      *kk = col - jl * a.nzc;
    } else {
      *jj = a.jj0 + o;
      *kk = col;
    }
  }
  // register slot of the inverse transform's input that holds forward output slot q
  // (last pass of radix R: slot q = (b, r) holds point jt + b*T + r*N/R = jt + (b + r*8/R)*T)
  static constexpr int RL = pass_radix(N, num_passes(N) - 1);
  static constexpr int inv_slot(int q) { return q / RL + (q % RL) * (8 / RL); }

  template <int PH>
  FG_HD static void phase(Regs& r, int block, int tid, double* lds, const XFusedArgs& a) {
    const LdsMap L = {C, 1, PN * C};
    constexpr int TR = PH / NPL;   // transform number: 0..NC-1 forward comp TR, NC..2NC-1 inverse comp TR-NC
    constexpr int LP = PH % NPL;   // phase inside the transform
    constexpr int comp = TR % NC;
    const cplx* tw = TW_LDS ? reinterpret_cast<const cplx*>(lds + TW_OFF) : a.tw;
    if (TW_LDS && PH == 0) {
      cplx* dst = reinterpret_cast<cplx*>(lds + TW_OFF);
      for (int i = tid; i < tw_total(N); i += THREADS) dst[i] = a.tw[i];   // fenced by the barrier after phase 0
    }
    if (PINGPONG) lds += ((LP / 2) % 2) * BUF_DOUBLES;
    if (PH == 0) {
      r.t = tid % C;
      r.jt = tid / C;
      locate(block, r.t, a, &r.base, &r.jj, &r.kk, &r.valid);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
        {
          const int j = Line<N>::first_index(r.jt, q);
          const long off = (long)j * a.ls + (XSPLIT ? (long)(j >> a.xsplit) * a.xjump : 0L);
          if constexpr (NTC >= 0) {
            const long lbase = r.valid ? r.base : r.base - r.t;
            r.v[c][q] = cload_stream(&a.data[c * a.comp_stride + lbase + off], NTC & 3);
          } else {
            r.v[c][q] = r.valid ? cload_stream(&a.data[c * a.comp_stride + r.base + off], a.nt) : cmake(0.0, 0.0);
          }
        }
      }
#if defined(__HIP_DEVICE_COMPILE__)
      // (NTC & 16: all loads of the tile are issued before anything else -- the scheduler otherwise sinks the loads of the
      // second and third component behind the first pass of the first)
      if constexpr (NTC >= 0 && (NTC & 16) != 0) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (TR < NC) {
      Line<N>::template phase<-1, LP>(r.v[comp], r.jt, lds, L, r.t, tw);
      if constexpr (NC == 1) if (LP == NPL - 1) {
        // scalar Green operator on the spectrum in registers
        const bool live = r.valid && r.kk < a.nzf;
        const double kpm1 = live ? a.kpm[1][r.jj] : 1.0, kpm2 = live ? a.kpm[2][r.kk] : 1.0;
        const cplx rb = cconj(a.half_root[r.jt]);
        cplx w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kx = Line<N>::last_index(r.jt, q);
          cplx e = cscale(a.scale, r.v[0][q]);
          if (live) {
            if (kx == 0 && r.jj == 0 && r.kk == 0) {
              e = cmake(0.0, 0.0);
            } else {
              const double kpm0 = (rb.im * a.xq[q].re + rb.re * a.xq[q].im) * a.inv_h;
              const double norm_kp2 = kpm0 * kpm0 + kpm1 * kpm1 + kpm2 * kpm2;
              e = cscale(a.c10 / norm_kp2, e);
            }
          }
          w[q] = e;
        }
        r.v[0][inv_slot(0)] = w[0]; r.v[0][inv_slot(1)] = w[1]; r.v[0][inv_slot(2)] = w[2]; r.v[0][inv_slot(3)] = w[3];
        r.v[0][inv_slot(4)] = w[4]; r.v[0][inv_slot(5)] = w[5]; r.v[0][inv_slot(6)] = w[6]; r.v[0][inv_slot(7)] = w[7];
      }
      if constexpr (NC == 3) if (TR == 2 && LP == NPL - 1) {
        // all three spectra are in registers: slot q holds kx = last_index(jt, q)
        const bool live = r.valid && r.kk < a.nzf;
        const double kpm1 = live ? a.kpm[1][r.jj] : 1.0, kpm2 = live ? a.kpm[2][r.kk] : 1.0;
        const cplx kp1 = live ? a.kp[1][r.jj] : cmake(0.0, 0.0), kp2 = live ? a.kp[2][r.kk] : cmake(0.0, 0.0);
        const cplx rb = cconj(a.half_root[r.jt]);
        cplx w[3][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kx = Line<N>::last_index(r.jt, q);
          const cplx et = cmul(rb, a.xq[q]);          // e^{i theta}
          const double kpm0 = et.im * a.inv_h;
          cplx t0 = cscale(a.scale, r.v[0][q]), t1 = cscale(a.scale, r.v[1][q]), t2 = cscale(a.scale, r.v[2][q]);
          cplx e0 = t0, e1 = t1, e2 = t2;
          if (live) {
            if (kx == 0 && r.jj == 0 && r.kk == 0) {
              e0 = e1 = e2 = cmake(0.0, 0.0);   // zero frequency  F:19924-19926
            } else {
              g0_point_rcp(t0, t1, t2, kpm0, kpm1, kpm2, cscale(kpm0, et), kp1, kp2, a.c10, a.c20, &e0, &e1, &e2);
            }
          }
          w[0][q] = e0;
          w[1][q] = e1;
          w[2][q] = e2;
        }
        // re-order for the inverse transform's first pass (same point set, different slot order)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          r.v[c][inv_slot(0)] = w[c][0]; r.v[c][inv_slot(1)] = w[c][1]; r.v[c][inv_slot(2)] = w[c][2];
          r.v[c][inv_slot(3)] = w[c][3]; r.v[c][inv_slot(4)] = w[c][4]; r.v[c][inv_slot(5)] = w[c][5];
          r.v[c][inv_slot(6)] = w[c][6]; r.v[c][inv_slot(7)] = w[c][7];
        }
      }
    } else {
      Line<N>::template phase<+1, LP>(r.v[comp], r.jt, lds, L, r.t, tw);
      if (LP == NPL - 1 && r.valid) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int j = Line<N>::last_index(r.jt, q);
          const long off = (long)j * a.ls + (XSPLIT ? (long)(j >> a.xsplit) * a.xjump : 0L);
          cstore_stream(&a.data[comp * a.comp_stride + r.base + off], r.v[comp][q], NTC >= 0 ? (NTC & 3) : a.nt);
        }
      }
    }
  }
  // a barrier is needed between consecutive phases except at a transform boundary
  // (the last phase of a transform does not touch LDS and the previous gather is already fenced)
  static constexpr bool barrier_after(int ph) {
    return (ph % NPL) != NPL - 1 && (!PINGPONG || (ph % NPL) % 2 == 0);
  }
};

}  // namespace fft
}  // namespace fg
