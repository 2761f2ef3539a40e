// z and y transforms of one z-y plane in ONE kernel (small grids: the complex plane [ny][nz/2+1] fits the 160 KB of LDS).
//
// The separate passes move every component through the memory system twice per direction (r2c_z: read + write, c2c_y: read +
// write) and cost two launches; at 128^3 and below the fields sit in the Infinity Cache and the passes are bounded by
// launch ramps and cache bandwidth, not HBM.  Here one workgroup owns a plane of one component: rows -> registers -> z
// transform (exchanges through LDS) -> spectrum of all rows in LDS -> y transform of its kz columns -> store; the inverse
// runs the other way.  Same arithmetic as R2CKernel / StridedKernel / C2RKernel (Line<N>, r2c_split, c2r_merge): every line
// transform is the same sequence of butterflies -- in the host emulation (tests/emulate runs these very functions) the
// results equal those of the separate passes bit for bit, on the device to the rounding of the compiler's contraction choices.
//
// Thread maps (THREADS = NY * M / 8, M = nz / 2):
//   z side: row = tid / TZ, jt = tid % TZ   (TZ = M / 8 threads per row, lanes along the row: 128-byte segments per row)
//   y side: col = tid % M,  jt = tid / M    (TY = NY / 8 threads per column, lanes along kz: contiguous stores)
// and the Nyquist column kz = M as a second, nearly empty round of the y side (TY threads).
// LDS layouts (fg_fft_core.h LdsMap): z side {1, LZ, LZ * NY} with LZ = M + M / 8 + 2 (lines apart), y side
// {M + 1, 1, pad8(NY) * (M + 1)} ([padded y][kz], kz fastest); the two alias the same allocation, a barrier between uses.
#pragma once

#include "fg_fft_kernels.h"

namespace fg {
namespace fft {

struct PlaneArgs {
  double* data;      // component 0; plane p at data + p * plane_doubles (rows nzp doubles apart)
  long plane_doubles;
  int nzp;           // doubles per row = 2 * nzc
  const cplx* tw_z;  // pass twiddles of M
  const cplx* wz;    // e^{-2 pi i k / nz}, k = 0..M
  const cplx* tw_y;  // pass twiddles of NY
};

template <int NY, int M>
struct PlaneGeom {
  static constexpr int TZ = M / 8, TY = NY / 8;
  static constexpr int THREADS = NY * TZ;
  static_assert(THREADS == M * TY, "thread maps of the two sides must have the same size");
  static_assert(TZ <= 64 && 64 % TZ == 0 && TY <= 64, "a row's threads, and the Nyquist round, must sit in one wave");
  static constexpr int LZ = M + M / 8 + 2;
  static constexpr int PY = NY + NY / 8;
  static constexpr int LDS_Z = 2 * LZ * NY, LDS_Y = 2 * PY * (M + 1);
  static constexpr int LDS_DOUBLES = LDS_Z > LDS_Y ? LDS_Z : LDS_Y;
  static constexpr int NPZ = Line<M>::NPHASE, NPY = Line<NY>::NPHASE;
  FG_HD static LdsMap zmap() { return LdsMap{1, LZ, LZ * NY}; }
  FG_HD static LdsMap ymap() { return LdsMap{M + 1, 1, PY * (M + 1)}; }
};

// forward: real rows -> r2c along z -> c2c along y, in place (fftVector's first two passes, F:18481-18500)
template <int NY, int M>
struct ZYKernel {
  using G = PlaneGeom<NY, M>;
  static constexpr int THREADS = G::THREADS, LDS_DOUBLES = G::LDS_DOUBLES;
  static constexpr int NPZ = G::NPZ, NPY = G::NPY, TZ = G::TZ, TY = G::TY;
  static constexpr int PH_SPLIT = NPZ, PH_PUT = NPZ + 1, PH_GATHER_A = NPZ + 2, PH_A = NPZ + 3, PH_GATHER_B = PH_A + NPY,
                       PH_B = PH_GATHER_B + 1, NPHASE = PH_B + NPY;
  // Synchronisation after phase PH: 2 = workgroup barrier, 1 = wave-local fence.  The TZ threads of a row sit in one wave
  // (TZ divides 64), so the z side exchanges inside waves; the columns of the y side span waves; the Nyquist round is wave 0's.
  static constexpr int barrier_after(int PH) { return PH < PH_SPLIT ? 1 : (PH < PH_A + NPY - 1 ? 2 : 1); }
  struct Regs {
    cplx v[8];
    cplx xm;        // X[M] of the row (jt == 0)
    double* plane;
    int jt, row;    // z side
    int jy, col;    // y side
  };
  template <int PH>
  FG_HD static void phase(Regs& r, int block, int tid, double* lds, const PlaneArgs& a) {
    const LdsMap LZm = G::zmap(), LYm = G::ymap();
    if constexpr (PH < NPZ) {
      if (PH == 0) {
        r.plane = a.data + (long)block * a.plane_doubles;
        r.jt = tid % TZ;
        r.row = tid / TZ;
        r.jy = tid / M;
        r.col = tid % M;
        const cplx* in = reinterpret_cast<const cplx*>(r.plane + (long)r.row * a.nzp);
#pragma unroll
        for (int q = 0; q < 8; ++q) r.v[q] = in[Line<M>::first_index(r.jt, q)];
      }
      Line<M>::template phase<-1, PH>(r.v, r.jt, lds, LZm, r.row, a.tw_z);
      if (PH == NPZ - 1) {   // natural-order spectrum of the packed row
#pragma unroll
        for (int q = 0; q < 8; ++q) lds_put(lds, LZm, Line<M>::last_index(r.jt, q), r.row, r.v[q]);
      }
    } else if constexpr (PH == PH_SPLIT) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = r.jt + q * TZ;
        r.v[q] = r2c_split(lds_get(lds, LZm, k, r.row), lds_get(lds, LZm, (M - k) % M, r.row), a.wz[k]);
      }
      if (r.jt == 0) {
        const cplx z0 = lds_get(lds, LZm, 0, r.row);
        r.xm = r2c_split(z0, z0, a.wz[M]);   // k = M (Nyquist): Z[M] := Z[0]
      }
    } else if constexpr (PH == PH_PUT) {
#pragma unroll
      for (int q = 0; q < 8; ++q) lds_put(lds, LYm, r.row, r.jt + q * TZ, r.v[q]);
      if (r.jt == 0) lds_put(lds, LYm, r.row, M, r.xm);
    } else if constexpr (PH == PH_GATHER_A) {
#pragma unroll
      for (int q = 0; q < 8; ++q) r.v[q] = lds_get(lds, LYm, Line<NY>::first_index(r.jy, q), r.col);
    } else if constexpr (PH < PH_GATHER_B) {
      constexpr int LP = PH - PH_A;
      Line<NY>::template phase<-1, LP>(r.v, r.jy, lds, LYm, r.col, a.tw_y);
      if (LP == NPY - 1) store(r, r.jy, r.col, a);
    } else if constexpr (PH == PH_GATHER_B) {
      if (tid < TY) {
#pragma unroll
        for (int q = 0; q < 8; ++q) r.v[q] = lds_get(lds, LYm, Line<NY>::first_index(tid, q), M);
      }
    } else {
      constexpr int LP = PH - PH_B;
      if (tid < TY) {
        Line<NY>::template phase<-1, LP>(r.v, tid, lds, LYm, M, a.tw_y);
        if (LP == NPY - 1) store(r, tid, M, a);
      }
    }
  }
  FG_HD static void store(const Regs& r, int jy, int col, const PlaneArgs& a) {
    cplx* out = reinterpret_cast<cplx*>(r.plane);
    const int nzc = a.nzp / 2;
#pragma unroll
    for (int q = 0; q < 8; ++q) out[(long)Line<NY>::last_index(jy, q) * nzc + col] = r.v[q];
  }
};

// inverse: c2c^-1 along y -> c2r along z (unnormalised; imaginary parts of the DC and Nyquist bins ignored like FFTW's c2r), in place
template <int NY, int M>
struct YZKernel {
  using G = PlaneGeom<NY, M>;
  static constexpr int THREADS = G::THREADS, LDS_DOUBLES = G::LDS_DOUBLES;
  static constexpr int NPZ = G::NPZ, NPY = G::NPY, TZ = G::TZ, TY = G::TY;
  static constexpr int PH_B = NPY, PH_MERGE = 2 * NPY, PH_Z = 2 * NPY + 1, NPHASE = PH_Z + NPZ;
  // y side: workgroup barriers; the Nyquist round (wave 0 alone) wave-local; both sides of the merge: workgroup; z side: wave-local
  static constexpr int barrier_after(int PH) { return PH < NPY - 1 ? 2 : (PH < PH_MERGE - 1 ? 1 : (PH <= PH_MERGE ? 2 : 1)); }
  struct Regs {
    cplx v[8];
    double* plane;
    int jt, row;
    int jy, col;
  };
  template <int PH>
  FG_HD static void phase(Regs& r, int block, int tid, double* lds, const PlaneArgs& a) {
    const LdsMap LZm = G::zmap(), LYm = G::ymap();
    if constexpr (PH < PH_B) {
      if (PH == 0) {
        r.plane = a.data + (long)block * a.plane_doubles;
        r.jt = tid % TZ;
        r.row = tid / TZ;
        r.jy = tid / M;
        r.col = tid % M;
        load(r, r.jy, r.col, a);
      }
      Line<NY>::template phase<+1, PH>(r.v, r.jy, lds, LYm, r.col, a.tw_y);
      if (PH == NPY - 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) lds_put(lds, LYm, Line<NY>::last_index(r.jy, q), r.col, r.v[q]);
      }
    } else if constexpr (PH < PH_MERGE) {
      constexpr int LP = PH - PH_B;
      if (tid < TY) {
        if (LP == 0) load(r, tid, M, a);
        Line<NY>::template phase<+1, LP>(r.v, tid, lds, LYm, M, a.tw_y);
        if (LP == NPY - 1) {
#pragma unroll
          for (int q = 0; q < 8; ++q) lds_put(lds, LYm, Line<NY>::last_index(tid, q), M, r.v[q]);
        }
      }
    } else if constexpr (PH == PH_MERGE) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int m = Line<M>::first_index(r.jt, q);
        cplx xk = lds_get(lds, LYm, r.row, m), xmk = lds_get(lds, LYm, r.row, M - m);
        if (m == 0) { xk.im = 0.0; xmk.im = 0.0; }   // FFTW's c2r ignores the imaginary parts of the DC and Nyquist bins
        r.v[q] = c2r_merge(xk, xmk, a.wz[m]);
      }
    } else {
      constexpr int LP = PH - PH_Z;
      Line<M>::template phase<+1, LP>(r.v, r.jt, lds, LZm, r.row, a.tw_z);
      if (LP == NPZ - 1) {
        cplx* out = reinterpret_cast<cplx*>(r.plane + (long)r.row * a.nzp);
#pragma unroll
        for (int q = 0; q < 8; ++q) out[Line<M>::last_index(r.jt, q)] = r.v[q];
      }
    }
  }
  FG_HD static void load(Regs& r, int jy, int col, const PlaneArgs& a) {
    const cplx* in = reinterpret_cast<const cplx*>(r.plane);
    const int nzc = a.nzp / 2;
#pragma unroll
    for (int q = 0; q < 8; ++q) r.v[q] = in[(long)Line<NY>::first_index(jy, q) * nzc + col];
  }
};

}  // namespace fft
}  // namespace fg
