// Power-of-two Stockham FFT building blocks shared by the HIP kernels and the
// host-side emulation (tests/emulate).  float64 throughout.
//
// Replaces FFTW's fftw_plan_dft_r2c_3d / c2r_3d used by the reference
// (F:7203-7245): unnormalised transforms, e^{-i...} forward, half spectrum
// along z, in place in the padded layout [nx][ny][nzp] <-> [nx][ny][nzc].
//
// One line of N complex points is handled by T = N/8 threads, each holding 8
// points in registers.  A pass of radix R in {2,4,8} performs 8/R butterflies
// per thread; between passes the points are exchanged through LDS.  With
//   in_index (j,r) = j + r*N/R
//   out_index(j,r) = (j/Ns)*Ns*R + (j%Ns) + r*Ns        (Ns = product of earlier radices)
// every pass reads "thread-contiguous" and the last pass stores to
// j + r*N/R again, so global loads and stores are coalesced.
#pragma once

#include "fg_common.h"

namespace fg {
namespace fft {

constexpr double kSqrtHalf = 0.70710678118654752440084436210484903928;

// ---------------------------------------------------------------- butterflies
// Natural-order DFTs of size 2/4/8, DIR = -1 forward (e^{-2 pi i nk/N}), +1 inverse.
template <int DIR>
FG_HD cplx mul_i_dir(cplx a) {  // multiply by DIR*i
  return DIR < 0 ? cmul_mi(a) : cmul_pi(a);
}

template <int DIR>
FG_HD void dft2(cplx* v) {
  cplx a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}

template <int DIR>
FG_HD void dft4(cplx* v) {
  cplx t0 = cadd(v[0], v[2]);
  cplx t1 = csub(v[0], v[2]);
  cplx t2 = cadd(v[1], v[3]);
  cplx t3 = mul_i_dir<DIR>(csub(v[1], v[3]));
  v[0] = cadd(t0, t2);
  v[1] = cadd(t1, t3);
  v[2] = csub(t0, t2);
  v[3] = csub(t1, t3);
}

template <int DIR>
FG_HD void dft8(cplx* v) {
  cplx e[4] = {v[0], v[2], v[4], v[6]};
  cplx o[4] = {v[1], v[3], v[5], v[7]};
  dft4<DIR>(e);
  dft4<DIR>(o);
  // w8^1 = (1 + DIR*i)/sqrt2, w8^2 = DIR*i, w8^3 = (-1 + DIR*i)/sqrt2
  cplx o1 = cmake(kSqrtHalf * (o[1].re - DIR * o[1].im), kSqrtHalf * (o[1].im + DIR * o[1].re));
  cplx o2 = mul_i_dir<DIR>(o[2]);
  cplx o3 = cmake(kSqrtHalf * (-o[3].re - DIR * o[3].im), kSqrtHalf * (-o[3].im + DIR * o[3].re));
  v[0] = cadd(e[0], o[0]);
  v[4] = csub(e[0], o[0]);
  v[1] = cadd(e[1], o1);
  v[5] = csub(e[1], o1);
  v[2] = cadd(e[2], o2);
  v[6] = csub(e[2], o2);
  v[3] = cadd(e[3], o3);
  v[7] = csub(e[3], o3);
}

template <int R, int DIR>
FG_HD void dftR(cplx* v) {
  if (R == 2) dft2<DIR>(v);
  else if (R == 4) dft4<DIR>(v);
  else dft8<DIR>(v);
}

// ---------------------------------------------------------------- schedules
// Radix schedule of a power-of-two N in [8, 4096]: as many radix-8 passes as
// possible, then one radix-4 or radix-2 pass.
constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n / 2); }
constexpr bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
constexpr int num_passes(int N) { return (ilog2(N) + 2) / 3; }
constexpr int pass_radix(int N, int p) {
  return (p < ilog2(N) / 3) ? 8 : (ilog2(N) % 3 == 1 ? 2 : 4);
}
constexpr int pass_ns(int N, int p) { return p == 0 ? 1 : pass_ns(N, p - 1) * pass_radix(N, p - 1); }
// offset (in cplx) of pass p's twiddle block inside the per-N table; pass 0 has none
constexpr int tw_block(int N, int p) { return (pass_radix(N, p) - 1) * (N / pass_radix(N, p)); }
constexpr int tw_offset(int N, int p) { return p <= 1 ? 0 : tw_offset(N, p - 1) + tw_block(N, p - 1); }
constexpr int tw_total(int N) { return num_passes(N) <= 1 ? 1 : tw_offset(N, num_passes(N) - 1) + tw_block(N, num_passes(N) - 1); }

// ---------------------------------------------------------------- LDS layout
// One 8-byte pad after every 8 points keeps the radix-8 scatter (stride 8)
// off a single bank group.
FG_HD int pad8(int m) { return m + (m >> 3); }

// Two layouts (doubles):  phys(m, c) = pad8(m)*SM + c*SC
//  * contiguous lines (z pass): lanes run along j, lines are separate:  SM = 1, SC = line stride
//  * strided lines (y/x pass):  lanes run along the C columns of a tile: SM = C, SC = 1
struct LdsMap {
  int sm, sc;
  int im_off;  // offset of the imaginary plane
};

FG_HD void lds_put(double* lds, const LdsMap& L, int m, int c, cplx v) {
  int p = pad8(m) * L.sm + c * L.sc;
  lds[p] = v.re;
  lds[p + L.im_off] = v.im;
}
FG_HD cplx lds_get(const double* lds, const LdsMap& L, int m, int c) {
  int p = pad8(m) * L.sm + c * L.sc;
  return cmake(lds[p], lds[p + L.im_off]);
}

// ---------------------------------------------------------------- passes
template <int N, int P>
struct Pass {
  static constexpr int R = pass_radix(N, P);
  static constexpr int NS = pass_ns(N, P);
  static constexpr int T = N / 8;   // threads per line
  static constexpr int NB = 8 / R;  // butterflies per thread
  static constexpr int NBF = N / R; // butterflies per line
  FG_HD static int in_index(int jt, int b, int r) { return jt + b * T + r * NBF; }
  FG_HD static int out_index(int jt, int b, int r) {
    int j = jt + b * T;
    return (j / NS) * NS * R + (j % NS) + r * NS;
  }
  // twiddle + butterflies on the 8 register points; tw = table block of this pass
  template <int DIR>
  FG_HD static void compute(cplx* v, int jt, const cplx* tw) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (NS > 1) {
        int j = jt + b * T;
#pragma unroll
        for (int r = 1; r < R; ++r) {
          cplx w = tw[(r - 1) * NBF + j];
          if (DIR > 0) w = cconj(w);
          v[b * R + r] = cmul(v[b * R + r], w);
        }
      }
      dftR<R, DIR>(v + b * R);
    }
  }
  FG_HD static void to_lds(const cplx* v, int jt, double* lds, const LdsMap& L, int c) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < R; ++r) lds_put(lds, L, out_index(jt, b, r), c, v[b * R + r]);
  }
  FG_HD static void from_lds(cplx* v, int jt, const double* lds, const LdsMap& L, int c) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < R; ++r) v[b * R + r] = lds_get(lds, L, in_index(jt, b, r), c);
  }
};

// Whole line transform split into barrier-separated phases so that the device
// kernel (phases separated by __syncthreads) and the host emulation (phases
// run for all threads in turn) execute the very same code.
//
//   phase 0            : [v holds pass-0 inputs]  compute pass 0, scatter to LDS
//   phase 2k-1 (k>=1)  : gather pass-k inputs from LDS
//   phase 2k           : compute pass k, scatter to LDS (not for the last pass)
// After the last compute phase v holds the outputs for out_index of the last
// pass, i.e. point  jt + b*T + r*N/R_last  in  v[b*R_last + r].
template <int N>
struct Line {
  static constexpr int NP = num_passes(N);
  static constexpr int NPHASE = 2 * NP - 1;
  static constexpr int T = N / 8;

  template <int DIR, int PH>
  FG_HD static void phase(cplx* v, int jt, double* lds, const LdsMap& L, int c, const cplx* tw) {
    constexpr int P = (PH + 1) / 2;
    if (PH % 2 == 1) {
      Pass<N, P>::from_lds(v, jt, lds, L, c);
    } else {
      Pass<N, P>::template compute<DIR>(v, jt, tw + tw_offset(N, P));
      if (P + 1 < NP) Pass<N, P>::to_lds(v, jt, lds, L, c);
    }
  }
  // register slot -> line point, for the first-pass inputs and last-pass outputs
  FG_HD static int first_index(int jt, int q) {
    constexpr int R = pass_radix(N, 0);
    return Pass<N, 0>::in_index(jt, q / R, q % R);
  }
  FG_HD static int last_index(int jt, int q) {
    constexpr int R = pass_radix(N, NP - 1);
    return Pass<N, NP - 1>::out_index(jt, q / R, q % R);
  }
};

// ---------------------------------------------------------------- r2c / c2r glue
// Real line of length 2M packed as M complex z_m = x[2m] + i x[2m+1].
// Forward split:  X[k] = (Z[k] + conj Z[M-k])/2 - i/2 * w^k (Z[k] - conj Z[M-k]),  w = e^{-2 pi i/(2M)}
// k = 0..M with Z[M] := Z[0].
FG_HD cplx r2c_split(cplx zk, cplx zmk, cplx wk) {
  cplx e = cmake(0.5 * (zk.re + zmk.re), 0.5 * (zk.im - zmk.im));
  cplx d = cmake(0.5 * (zk.re - zmk.re), 0.5 * (zk.im + zmk.im));  // (Z[k] - conj Z[M-k])/2
  cplx o = cmul(wk, d);
  return cmake(e.re + o.im, e.im - o.re);  // e - i*o
}
// Inverse merge (unnormalised c2r):  Z'[k] = (X[k] + conj X[M-k]) + i conj(w^k) (X[k] - conj X[M-k])
FG_HD cplx c2r_merge(cplx xk, cplx xmk, cplx wk) {
  cplx e = cmake(xk.re + xmk.re, xk.im - xmk.im);
  cplx d = cmake(xk.re - xmk.re, xk.im + xmk.im);
  cplx o = cmul(cconj(wk), d);
  return cmake(e.re - o.im, e.im + o.re);  // e + i*o
}

}  // namespace fft
}  // namespace fg
