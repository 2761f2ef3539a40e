// Hand-written 3-D r2c/c2r FFT for gfx950 (no hipFFT/rocFFT).
//
// Three batched 1-D passes per transform, each in place:
//   z : rows of nz reals <-> nzc complex (packed-real trick, lanes along z => coalesced 16 B/lane)
//   y : lines nzc complex apart, tiles of C adjacent kz columns (C*16 B contiguous segments)
//   x : lines ny*nzc apart, tiles of C adjacent (ky,kz) columns
// Butterflies live in registers (8 points/thread), exchanges go through padded LDS.
// Axes that are not a power of two in [8,1024] use O(n^2) DFT kernels via a scratch
// component (correctness path for the odd grids of the reference's self tests).
#include "fg_fft.h"

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "fg_fft_kernels.h"
#include "fg_fft_plane.h"
#include "fg_fft_smooth.h"
#include "fg_fft_smooth_dev.h"
#include "fg_fft_tables.h"
#include "fg_hip_util.h"

namespace fg {

using namespace fft;

namespace {

// Kernels whose lines never span two waves (z passes: T threads per line, T divides 64) exchange through LDS inside
// one wave only: the LDS queue of a wave is in order, so a compiler-level fence replaces the workgroup barrier and
// the waves of a workgroup drift freely (loads of one overlap the butterflies of another).
template <class K>
struct WaveLocal {
  static constexpr bool value = false;
};
template <int M, int LINES, bool MIRROR>
struct WaveLocal<R2CKernel<M, LINES, MIRROR>> {
  static constexpr bool value = (M / 8) <= 64 && 64 % (M / 8) == 0;
};
template <int M, int LINES, bool MIRROR>
struct WaveLocal<C2RKernel<M, LINES, MIRROR>> {
  static constexpr bool value = (M / 8) <= 64 && 64 % (M / 8) == 0;
};

template <class K, class Args, int PH>
struct DevicePhases {
  __device__ __forceinline__ static void run(typename K::Regs& r, int block, int tid, double* lds, const Args& a) {
    K::template phase<PH>(r, block, tid, lds, a);
    if constexpr (PH + 1 < K::NPHASE) {
      if constexpr (WaveLocal<K>::value) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      } else {
        __syncthreads();
      }
      DevicePhases<K, Args, PH + 1>::run(r, block, tid, lds, a);
    }
  }
};

template <class K>
__global__ __launch_bounds__(K::THREADS) void k_strided(StridedArgs a, long comp_stride) {
  extern __shared__ __align__(16) double lds[];
  a.data += (long)blockIdx.y * comp_stride;
  if (a.out) a.out += (long)blockIdx.y * a.out_cs;
  typename K::Regs r;
  int b = blockIdx.x;
  if (a.xcd_order && gridDim.x % 8 == 0) b = (b % 8) * (gridDim.x / 8) + b / 8;
  DevicePhases<K, StridedArgs, 0>::run(r, b, threadIdx.x, lds, a);
}

template <class K>
__global__ __launch_bounds__(K::THREADS) void k_zpass(ZArgs a, long comp_stride) {
  extern __shared__ __align__(16) double lds[];
  a.data += (long)blockIdx.y * comp_stride;
  typename K::Regs r;
  DevicePhases<K, ZArgs, 0>::run(r, blockIdx.x, threadIdx.x, lds, a);
}

// tools/xfused_probe.hip compiles this file with -DFG_PROBE: cycle stamps (s_memtime) of one wave per sampled
// workgroup at every phase boundary of the fused x pass, to see where a tile's time goes.  Empty in the library.
#ifdef FG_PROBE
#ifndef FG_PROBE_STRIDE
#define FG_PROBE_STRIDE 256
#endif
constexpr int kProbeBlocks = 64, kProbeSlots = 40, kProbeStride = FG_PROBE_STRIDE;
__device__ unsigned long long g_probe[kProbeBlocks][kProbeSlots];
#define FG_PROBE_MARK(ph)                                                                        \
  do {                                                                                           \
    if (tid == FG_PROBE_THREAD && block % kProbeStride == 0 && block / kProbeStride < kProbeBlocks) \
      g_probe[block / kProbeStride][(ph)] = __builtin_readcyclecounter();                        \
  } while (0)
#else
#define FG_PROBE_MARK(ph) do { } while (0)
#endif

// z + y transforms of one plane per workgroup (fg_fft_plane.h); blockIdx.y = component
template <class K, int PH>
struct DevicePhasesP {
  __device__ __forceinline__ static void run(typename K::Regs& r, int block, int tid, double* lds, const PlaneArgs& a) {
    FG_PROBE_MARK(PH);
    K::template phase<PH>(r, block, tid, lds, a);
    if constexpr (PH + 1 == K::NPHASE) FG_PROBE_MARK(PH + 1);
    if constexpr (PH + 1 < K::NPHASE) {
      if constexpr (K::barrier_after(PH) == 2 || K::THREADS <= 64) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      DevicePhasesP<K, PH + 1>::run(r, block, tid, lds, a);
    }
  }
};
template <class K>
__global__ __launch_bounds__(K::THREADS) void k_plane(PlaneArgs a, long comp_stride) {
  extern __shared__ __align__(16) double lds[];
  a.data += (long)blockIdx.y * comp_stride;
  typename K::Regs r;
  DevicePhasesP<K, 0>::run(r, blockIdx.x, threadIdx.x, lds, a);
}

template <class K, int PH>
struct DevicePhasesX {
  __device__ __forceinline__ static void run(typename K::Regs& r, int block, int tid, double* lds, const XFusedArgs& a) {
    FG_PROBE_MARK(PH);
    K::template phase<PH>(r, block, tid, lds, a);
    if constexpr (PH + 1 < K::NPHASE) {
      if constexpr (K::barrier_after(PH)) __syncthreads();
      DevicePhasesX<K, PH + 1>::run(r, block, tid, lds, a);
    } else {
      FG_PROBE_MARK(PH + 1);
    }
  }
};

template <class K>
__global__ __launch_bounds__(K::THREADS) void k_xfused(XFusedArgs a) {
  extern __shared__ __align__(16) double lds[];
  typename K::Regs r;
  // workgroups are dealt round-robin to the eight XCDs: give every XCD a contiguous run of tiles (512^3: 1.84 -> 1.80 ms,
  // 256^3: 0.200 -> 0.188 ms against neighbouring tiles on different XCDs)
  int b = blockIdx.x;
  if (a.xcd_order && gridDim.x % 8 == 0) b = (b % 8) * (gridDim.x / 8) + b / 8;
  DevicePhasesX<K, 0>::run(r, b, threadIdx.x, lds, a);
}

__global__ void k_dft_strided_generic(const cplx* src, cplx* dst, long ls, long os, int ncols, int nouter, int n,
                                      int dir, double scale, const cplx* w) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)nouter * n * ncols;
  if (idx >= total) return;
  int col = idx % ncols;
  long rest = idx / ncols;
  int k = rest % n;
  int o = rest / n;
  dft_strided_point(src, dst, (long)o * os + col, ls, n, k, dir, scale, w);
}

__global__ void k_r2c_generic(const double* src, double* dst, long nrows, int nz, int nzp, const cplx* w) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int nzc = nz / 2 + 1;
  if (idx >= nrows * nzc) return;
  long row = idx / nzc;
  int k = idx % nzc;
  r2c_point(src + row * nzp, reinterpret_cast<cplx*>(dst + row * nzp), nz, k, w);
}

__global__ void k_c2r_generic(const double* src, double* dst, long nrows, int nz, int nzp, const cplx* w) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nrows * nz) return;
  long row = idx / nz;
  int m = idx % nz;
  c2r_point(reinterpret_cast<const cplx*>(src + row * nzp), dst + row * nzp, nz, m, w);
}

__global__ void k_scale(double* x, long n, double s) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

// lengths with factors 2, 3, 5, 7, 11, 13: the Stockham tile kernels live in fg_fft_smooth_yz.hip / fg_fft_smooth_x.hip
// (translation units of their own: their many instances compile beside this file)

int nt_loads_env() {
  return 15;   // streaming loads in r2c, the strided passes, the fused x pass and the mirrored c2r (bits 1, 2, 4, 8)
}

bool fast_len(int n) { return is_pow2(n) && n >= 8 && n <= 1024; }

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"   // the run-time-p instances keep their loops rolled
// ---------------------------------------------------------------- lengths p * 2^k, p odd <= 25
// N = p M: the line is p interleaved sub-lines x_r[m] = x[p m + r] of the power-of-two length M.  Forward: the M-point
// kernels transform the sub-lines in place (line stride p * ls), Y_r[k] then sits at point p k + r, and one combine
// sweep forms  X[k + M s] = sum_r w_N^{r k} w_p^{r s} Y_r[k]  (out of place, through the scratch component).
// Inverse: the combine sweep first,  Z_r[k] = conj(w_N^{r k}) sum_s X[k + M s] conj(w_p^{r s})  to point p k + r,
// then the M-point inverse kernels on the sub-lines.  3x the traffic of a native pass instead of an O(N^2) DFT.
constexpr int kMaxOddFactor = 25;   // 25: the decimal sizes 200, 400, 800

int mixed_factor(int n) {   // p if n = p * 2^k with 2^k a fast length, else 0
  for (int p = 3; p <= kMaxOddFactor; p += 2)
    if (n % p == 0 && fast_len(n / p)) return p;
  return 0;
}



// w: e^{-2 pi i j / N}, j < N
template <int DIR, int P>   // P = 0: run-time p (indexed local arrays); 3, 5, 7: unrolled, everything in registers
__global__ __launch_bounds__(256) void k_mixed_combine(const cplx* src, cplx* dst, long ls, long os, int ncols, int nouter,
                                                       int M, int p_rt, double scale, const cplx* w) {
  const int p = P ? P : p_rt;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)nouter * M * ncols;
  if (idx >= total) return;
  const int col = (int)(idx % ncols);
  const long rest = idx / ncols;
  const int k = (int)(rest % M);
  const long base = (rest / M) * os + col;
  const int N = p * M;
  cplx in[P ? P : kMaxOddFactor];
#pragma unroll
  for (int r = 0; r < p; ++r) {
    // forward: Y_r[k] at point p k + r;  inverse: X[k + M r] at its natural point
    const long pt = DIR < 0 ? (long)p * k + r : (long)k + (long)M * r;
    in[r] = src[base + pt * ls];
  }
#pragma unroll
  for (int o = 0; o < p; ++o) {
    cplx acc = cmake(0.0, 0.0);
    // the table index r * step mod N advances by step with one conditional subtraction (no division)
    if (DIR < 0) {   // output s = o:  sum_r w_N^{r (k + M s)} Y_r
      const int step = k + M * o;   // < N
      int wi = 0;
#pragma unroll
      for (int r = 0; r < p; ++r) {
        acc = cadd(acc, cmul(in[r], w[wi]));
        wi += step;
        if (wi >= N) wi -= N;
      }
      dst[base + ((long)k + (long)M * o) * ls] = cscale(scale, acc);
    } else {         // output r = o:  conj(w_N^{r k}) sum_s X_s conj(w_p^{r s})
      const int step = (int)(((long)o * M) % N);
      int wi = 0;
#pragma unroll
      for (int sidx = 0; sidx < p; ++sidx) {
        acc = cadd(acc, cmul(in[sidx], cconj(w[wi])));
        wi += step;
        if (wi >= N) wi -= N;
      }
      acc = cmul(acc, cconj(w[o * k]));   // o k < N
      dst[base + ((long)p * k + o) * ls] = cscale(scale, acc);
    }
  }
}

// The same combine in place: a workgroup owns whole lines (a tile of 8 adjacent columns x all N points), computes every
// output of the tile into LDS, and only after a barrier writes the tile back -- no scratch component, no copy
// (2x instead of 3x the traffic of a native pass).  N * 128 bytes of LDS: used while that fits (N <= 1152).
template <int DIR, int P>
__global__ __launch_bounds__(256) void k_mixed_combine_tile(cplx* data, long ls, long os, int ncols, int tiles_per_outer, int M,
                                                            int p_rt, double scale, const cplx* w) {
  extern __shared__ __align__(16) double lds_raw[];
  cplx* img = reinterpret_cast<cplx*>(lds_raw);   // [N][8]
  const int p = P ? P : p_rt;
  const int N = p * M;
  const int o = blockIdx.x / tiles_per_outer;
  const int col0 = (blockIdx.x % tiles_per_outer) * 8;
  const int t = threadIdx.x & 7;
  const bool valid = col0 + t < ncols;
  const long base = (long)o * os + col0 + t;
  for (int k = (int)(threadIdx.x >> 3); k < M; k += 32) {
    cplx in[P ? P : kMaxOddFactor];
#pragma unroll
    for (int r = 0; r < p; ++r) {
      const long pt = DIR < 0 ? (long)p * k + r : (long)k + (long)M * r;
      in[r] = valid ? data[base + pt * ls] : cmake(0.0, 0.0);
    }
#pragma unroll
    for (int q = 0; q < p; ++q) {
      cplx acc = cmake(0.0, 0.0);
      if (DIR < 0) {
        const int step = k + M * q;
        int idx = 0;
#pragma unroll
        for (int r = 0; r < p; ++r) {
          acc = cadd(acc, cmul(in[r], w[idx]));
          idx += step;
          if (idx >= N) idx -= N;
        }
        img[(k + M * q) * 8 + t] = cscale(scale, acc);
      } else {
        const int step = (int)(((long)q * M) % N);
        int idx = 0;
#pragma unroll
        for (int sidx = 0; sidx < p; ++sidx) {
          acc = cadd(acc, cmul(in[sidx], cconj(w[idx])));
          idx += step;
          if (idx >= N) idx -= N;
        }
        img[(p * k + q) * 8 + t] = cscale(scale, cmul(acc, cconj(w[q * k])));
      }
    }
  }
  __syncthreads();
  if (valid)
    for (int pt = (int)(threadIdx.x >> 3); pt < N; pt += 32) data[base + (long)pt * ls] = img[pt * 8 + t];
}

// One kernel for the whole pass where the tile fits (p = 3, 5, 7, 9; N = p M <= 1024 threads, exchange planes <= 144 KB):
// the workgroup holds an 8-column tile of whole lines, thread (jt, r, t) runs the M-point transform of sub-line r of
// column t with the phase code of the power-of-two kernels (the p * 8 sub-lines are just more LDS columns), and the
// combine step goes through an LDS image of the tile -- one read and one write of the data, like a native pass.
template <int M, int DIR, int PH>
struct MixedPhases {
  __device__ __forceinline__ static void run(cplx* v, int jt, double* lds, const fft::LdsMap& L, int c, const cplx* tw) {
    fft::Line<M>::template phase<DIR, PH>(v, jt, lds, L, c, tw);
    if constexpr (PH + 1 < fft::Line<M>::NPHASE) {
      __syncthreads();
      MixedPhases<M, DIR, PH + 1>::run(v, jt, lds, L, c, tw);
    }
  }
};

template <int M, int P, int DIR>
__global__ __launch_bounds__(M * P) void k_strided_mixed(StridedArgs a, long comp_stride, const cplx* wN) {
  using namespace fft;
  constexpr int T = M / 8, COLS = P * 8, THREADS = T * COLS, PN = M + M / 8, N = P * M;
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);   // [P][M][8], aliases the exchange planes (never live together)
  a.data += (long)blockIdx.y * comp_stride;
  const int tid = threadIdx.x;
  const int cp = tid % COLS, jt = tid / COLS, r = cp / 8, t = cp % 8;
  const int o = blockIdx.x / a.tiles_per_outer;
  const int col = (blockIdx.x % a.tiles_per_outer) * 8 + t;
  const bool valid = col < a.ncols;
  const long base = (long)o * a.os + (valid ? col : 0);
  const LdsMap L = {COLS, 1, PN * COLS};
  cplx v[8];
  auto combine = [&](auto&& get, auto&& put) {   // thread (k, t): p inputs -> p outputs
    for (int k = tid / 8; k < M; k += THREADS / 8) {
      cplx in[P];
#pragma unroll
      for (int q = 0; q < P; ++q) in[q] = get(k, q);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        cplx acc = cmake(0.0, 0.0);
        if (DIR < 0) {   // X[k + M q] = sum_r w_N^{r (k + M q)} Y_r[k]
          const int step = k + M * q;
          int idx = 0;
#pragma unroll
          for (int rr = 0; rr < P; ++rr) {
            acc = cadd(acc, cmul(in[rr], wN[idx]));
            idx += step;
            if (idx >= N) idx -= N;
          }
        } else {         // Z_q[k] = conj(w_N^{q k}) sum_s X[k + M s] conj(w_p^{q s})
          const int step = (q * M) % N;
          int idx = 0;
#pragma unroll
          for (int sidx = 0; sidx < P; ++sidx) {
            acc = cadd(acc, cmul(in[sidx], cconj(wN[idx])));
            idx += step;
            if (idx >= N) idx -= N;
          }
          acc = cmul(acc, cconj(wN[q * k]));
        }
        put(k, q, acc);
      }
    }
  };
  if (DIR < 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      v[q] = valid ? cload_stream(&a.data[base + ((long)P * Line<M>::first_index(jt, q) + r) * a.ls], a.nt) : cmake(0.0, 0.0);
  } else {
    // inverse: combine first, from memory into the image [r][k][t]
    combine([&](int k, int q) { return valid ? cload_stream(&a.data[base + ((long)k + (long)M * q) * a.ls], a.nt) : cmake(0.0, 0.0); },
            [&](int k, int q, cplx z) { img[((long)q * M + k) * 8 + t] = z; });
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = img[((long)r * M + Line<M>::first_index(jt, q)) * 8 + t];
    __syncthreads();   // the image is overwritten by the exchange planes from here on
  }
  // the M-point transform of this thread's sub-line (barrier between phases, as in k_strided)
  MixedPhases<M, DIR, 0>::run(v, jt, lds, L, cp, a.tw);
  if (DIR < 0) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) img[((long)r * M + Line<M>::last_index(jt, q)) * 8 + t] = v[q];
    __syncthreads();
    combine([&](int k, int q) { return img[((long)q * M + k) * 8 + t]; },
            [&](int k, int q, cplx z) {
              if (valid) cstore_stream(&a.data[base + ((long)k + (long)M * q) * a.ls], cscale(a.scale, z), a.nt);
            });
  } else if (valid) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      cstore_stream(&a.data[base + ((long)P * Line<M>::last_index(jt, q) + r) * a.ls], cscale(a.scale, v[q]), a.nt);
  }
}

template <int M, int P>
bool launch_strided_mixed(const StridedArgs& a0, int nouter, int dir, int ncomp, long cs, const cplx* wN, hipStream_t s) {
  constexpr int PN = M + M / 8;
  constexpr size_t lds = 2 * PN * P * 8 * sizeof(double);
  if constexpr (M * P > 1024 || lds > 144 * 1024) {
    return false;
  } else {
    StridedArgs a = a0;
    a.tiles_per_outer = (a.ncols + 7) / 8;
    const dim3 grid((unsigned)((long)a.tiles_per_outer * nouter), ncomp);
    static PerDeviceOnce configured;
    if (auto once = configured.first_use()) {
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_strided_mixed<M, P, -1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_strided_mixed<M, P, +1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (dir < 0) hipLaunchKernelGGL((k_strided_mixed<M, P, -1>), grid, dim3(M * P), lds, s, a, cs, wN);
    else hipLaunchKernelGGL((k_strided_mixed<M, P, +1>), grid, dim3(M * P), lds, s, a, cs, wN);
    FG_HIP_CHECK(hipGetLastError());
    return true;
  }
}

// The z passes in one kernel each, same construction: a workgroup owns LINES packed rows, thread (l, jt, r) runs the
// MP-point transform of sub-row r of row l (lanes run over (jt, r): the loads p * idx + r are contiguous), the
// combine + real split (r2c) / merge + inverse combine (c2r) goes through an LDS image [l][r][k'].
template <int MP, int P, int LINES, bool FWD>
__global__ __launch_bounds__((MP / 8) * P * LINES) void k_z_mixed(double* data, long nrows, int nzp, long comp_stride,
                                                                 const cplx* tw, const cplx* wn) {
  using namespace fft;
  constexpr int T = MP / 8, COLS = P * LINES, THREADS = T * COLS, PN = MP + MP / 8, M = P * MP, nz = 2 * M;
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);   // [LINES][P][MP]
  data += (long)blockIdx.y * comp_stride;
  const int tid = threadIdx.x;
  // lanes: r fastest, then jt, then the row: global accesses p * idx + r are contiguous along (jt, r)
  const int r = tid % P, jt = (tid / P) % T, l = tid / (P * T);
  const int cp = l * P + r;   // LDS column of this sub-row
  const long row0 = (long)blockIdx.x * LINES;
  const bool valid = row0 + l < nrows;
  cplx* row = reinterpret_cast<cplx*>(data + (row0 + (valid ? l : 0)) * nzp);
  const LdsMap L = {COLS, 1, PN * COLS};
  cplx v[8];
  if (FWD) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = valid ? row[P * Line<MP>::first_index(jt, q) + r] : cmake(0.0, 0.0);
    MixedPhases<MP, -1, 0>::run(v, jt, lds, L, cp, tw);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) img[((long)l * P + r) * MP + Line<MP>::last_index(jt, q)] = v[q];
    __syncthreads();
    // combine + split: item (row, k <= M/2) -> X[k], X[M - k]
    constexpr int half = M / 2 + 1;
    for (int it = tid; it < LINES * half; it += THREADS) {
      const int lr = it / half, k = it % half;
      if (row0 + lr >= nrows) break;
      const cplx* in = img + (long)lr * P * MP;
      auto zfull = [&](int K) {
        const int kk = K % MP, step = 2 * K;
        cplx acc = cmake(0.0, 0.0);
        int idx = 0;
#pragma unroll
        for (int rr = 0; rr < P; ++rr) {
          acc = cadd(acc, cmul(in[rr * MP + kk], wn[idx]));
          idx += step;
          if (idx >= nz) idx -= nz;
        }
        return acc;
      };
      const cplx zk = zfull(k), zm = zfull((M - k) % M);
      cplx* out = reinterpret_cast<cplx*>(data + (row0 + lr) * nzp);
      out[k] = r2c_split(zk, zm, wn[k]);
      out[M - k] = r2c_split(zm, zk, wn[M - k]);
    }
  } else {
    // merge + inverse combine: item (row, k' < MP) -> Z_r[k'] for all r
    for (int it = tid; it < LINES * MP; it += THREADS) {
      const int lr = it / MP, kk = it % MP;
      if (row0 + lr >= nrows) break;
      const cplx* in = reinterpret_cast<const cplx*>(data + (row0 + lr) * nzp);
      cplx z[P];
#pragma unroll
      for (int sidx = 0; sidx < P; ++sidx) {
        const int K = kk + MP * sidx;
        cplx xk = in[K], xm = in[M - K];
        if (K == 0) { xk.im = 0.0; xm.im = 0.0; }
        z[sidx] = c2r_merge(xk, xm, wn[K]);
      }
#pragma unroll
      for (int rr = 0; rr < P; ++rr) {
        cplx acc = cmake(0.0, 0.0);
        const int step = (2 * rr * MP) % nz;
        int idx = 0;
#pragma unroll
        for (int sidx = 0; sidx < P; ++sidx) {
          acc = cadd(acc, cmul(z[sidx], cconj(wn[idx])));
          idx += step;
          if (idx >= nz) idx -= nz;
        }
        img[((long)lr * P + rr) * MP + kk] = cmul(acc, cconj(wn[2 * rr * kk]));
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = img[((long)l * P + r) * MP + Line<MP>::first_index(jt, q)];
    __syncthreads();
    MixedPhases<MP, +1, 0>::run(v, jt, lds, L, cp, tw);
    if (valid) {
#pragma unroll
      for (int q = 0; q < 8; ++q) row[P * Line<MP>::last_index(jt, q) + r] = v[q];
    }
  }
}

template <int MP, int P>
bool launch_z_mixed(double* data, long nrows, int nzp, int ncomp, long comp_stride, bool fwd, const cplx* tw, const cplx* wn,
                    hipStream_t s) {
  constexpr int LINES = (2048 / (MP * P)) > 1 ? (2048 / (MP * P)) : 1;
  constexpr int THREADS = (MP / 8) * P * LINES;
  constexpr int PN = MP + MP / 8;
  constexpr size_t ex = 2 * PN * P * LINES * sizeof(double), im = (size_t)LINES * P * MP * sizeof(cplx);
  constexpr size_t lds = ex > im ? ex : im;
  if constexpr (THREADS > 1024 || lds > 144 * 1024) {
    return false;
  } else {
    const dim3 grid((unsigned)((nrows + LINES - 1) / LINES), ncomp);
    static PerDeviceOnce configured;
    if (auto once = configured.first_use()) {
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_z_mixed<MP, P, LINES, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_z_mixed<MP, P, LINES, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (fwd) hipLaunchKernelGGL((k_z_mixed<MP, P, LINES, true>), grid, dim3(THREADS), lds, s, data, nrows, nzp, comp_stride, tw, wn);
    else hipLaunchKernelGGL((k_z_mixed<MP, P, LINES, false>), grid, dim3(THREADS), lds, s, data, nrows, nzp, comp_stride, tw, wn);
    FG_HIP_CHECK(hipGetLastError());
    return true;
  }
}

template <int P>
bool z_mixed_p(int mp, double* data, long nrows, int nzp, int ncomp, long cs, bool fwd, const cplx* tw, const cplx* wn,
               hipStream_t s) {
  switch (mp) {
    case 8: return launch_z_mixed<8, P>(data, nrows, nzp, ncomp, cs, fwd, tw, wn, s);
    case 16: return launch_z_mixed<16, P>(data, nrows, nzp, ncomp, cs, fwd, tw, wn, s);
    case 32: return launch_z_mixed<32, P>(data, nrows, nzp, ncomp, cs, fwd, tw, wn, s);
    case 64: return launch_z_mixed<64, P>(data, nrows, nzp, ncomp, cs, fwd, tw, wn, s);
    case 128: return launch_z_mixed<128, P>(data, nrows, nzp, ncomp, cs, fwd, tw, wn, s);
    case 256: return launch_z_mixed<256, P>(data, nrows, nzp, ncomp, cs, fwd, tw, wn, s);
    default: return false;
  }
}

// The fused pass (x-FFT, 1/N, Green operator, x-iFFT on three components) for N = p M, p = 3 or 5: per component the
// sub-line transforms and the LDS image as in k_strided_mixed; the combine outputs X[k + M s] of a (k, column) item stay
// in the registers of the thread that formed them -- the same thread owns every input of the inverse combine of that
// item, so the Green operator and the inverse combine need no exchange.  One read and one write of the three spectra.
template <int M, int P>
__global__ __launch_bounds__(M * P) void k_xfused_mixed(XFusedArgs a, const cplx* wN) {
  using namespace fft;
  constexpr int T = M / 8, COLS = P * 8, THREADS = T * COLS, PN = M + M / 8, N = P * M;
  constexpr int ITEMS = (8 + P - 1) / P;   // (k, column) items per thread: M * 8 items over M * P threads
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  const int tid = threadIdx.x;
  const int cp = tid % COLS, jt = tid / COLS, r = cp / 8, t = cp % 8;
  const int o = blockIdx.x / a.tiles_per_outer;
  const int col = (blockIdx.x % a.tiles_per_outer) * 8 + t;
  const bool valid = col < a.ncols;
  const int colc = valid ? col : a.ncols - 1;
  const long base = (long)o * a.os + colc;
  int jj, kk;
  if (a.flat_cols) {
    const int jl = colc / a.nzc;
    jj = a.jj0 + jl;
    kk = colc - jl * a.nzc;
  } else {
    jj = a.jj0 + o;
    kk = colc;
  }
  const LdsMap L = {COLS, 1, PN * COLS};
  cplx v[8];
  cplx X[ITEMS][3][P];
  // ---- forward: three components
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const cplx* src = a.data + c * a.comp_stride;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = cload_stream(&src[base + ((long)P * Line<M>::first_index(jt, q) + r) * a.ls], a.nt);
    MixedPhases<M, -1, 0>::run(v, jt, lds, L, cp, a.tw);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) img[((long)r * M + Line<M>::last_index(jt, q)) * 8 + t] = v[q];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int k = tid / 8 + i * (THREADS / 8);
      if (k < M) {
        cplx in[P];
#pragma unroll
        for (int rr = 0; rr < P; ++rr) in[rr] = img[((long)rr * M + k) * 8 + t];
#pragma unroll
        for (int sidx = 0; sidx < P; ++sidx) {
          cplx acc = cmake(0.0, 0.0);
          const int step = k + M * sidx;
          int idx = 0;
#pragma unroll
          for (int rr = 0; rr < P; ++rr) {
            acc = cadd(acc, cmul(in[rr], wN[idx]));
            idx += step;
            if (idx >= N) idx -= N;
          }
          X[i][c][sidx] = cscale(a.scale, acc);
        }
      }
    }
    __syncthreads();   // the image is overwritten by the next component's exchange planes
  }
  // ---- Green operator and inverse combine, in registers
  const bool live = valid && kk < a.nzf;
  const double kpm1 = a.kpm[1][jj], kpm2 = a.kpm[2][live ? kk : 0];
  const cplx kp1 = a.kp[1][jj], kp2 = a.kp[2][live ? kk : 0];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int k = tid / 8 + i * (THREADS / 8);
    if (k < M) {
#pragma unroll
      for (int sidx = 0; sidx < P; ++sidx) {
        const int kx = k + M * sidx;
        const cplx t0 = X[i][0][sidx], t1 = X[i][1][sidx], t2 = X[i][2][sidx];
        if (live) {
          cplx e0, e1, e2;
          g0_point_rcp(t0, t1, t2, a.kpm[0][kx], kpm1, kpm2, a.kp[0][kx], kp1, kp2, a.c10, a.c20, &e0, &e1, &e2);
          const bool zero = kx == 0 && jj == 0 && kk == 0;   // zero frequency  F:19924-19926
          X[i][0][sidx] = zero ? cmake(0.0, 0.0) : e0;
          X[i][1][sidx] = zero ? cmake(0.0, 0.0) : e1;
          X[i][2][sidx] = zero ? cmake(0.0, 0.0) : e2;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        cplx in[P];
#pragma unroll
        for (int sidx = 0; sidx < P; ++sidx) in[sidx] = X[i][c][sidx];
#pragma unroll
        for (int rr = 0; rr < P; ++rr) {   // Z_r[k] = conj(w_N^{r k}) sum_s X[k + M s] conj(w_p^{r s})
          cplx acc = cmake(0.0, 0.0);
          const int step = (rr * M) % N;
          int idx = 0;
#pragma unroll
          for (int sidx = 0; sidx < P; ++sidx) {
            acc = cadd(acc, cmul(in[sidx], cconj(wN[idx])));
            idx += step;
            if (idx >= N) idx -= N;
          }
          X[i][c][rr] = cmul(acc, cconj(wN[rr * k]));
        }
      }
    }
  }
  // ---- inverse: three components
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int k = tid / 8 + i * (THREADS / 8);
      if (k < M) {
#pragma unroll
        for (int rr = 0; rr < P; ++rr) img[((long)rr * M + k) * 8 + t] = X[i][c][rr];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = img[((long)r * M + Line<M>::first_index(jt, q)) * 8 + t];
    __syncthreads();
    MixedPhases<M, +1, 0>::run(v, jt, lds, L, cp, a.tw);
    if (valid) {
      cplx* dst = a.data + c * a.comp_stride;
#pragma unroll
      for (int q = 0; q < 8; ++q) cstore_stream(&dst[base + ((long)P * Line<M>::last_index(jt, q) + r) * a.ls], v[q], a.nt);
    }
    __syncthreads();   // exchange planes -> image of the next component
  }
}

template <int M, int P>
bool launch_xfused_mixed(XFusedArgs a, int nouter, const cplx* wN, hipStream_t s, bool probe_only) {
  constexpr int PN = M + M / 8;
  constexpr size_t lds = 2 * PN * P * 8 * sizeof(double);
  // N <= 512 like the power-of-two fused pass: 640 and 768 need 0.9-1.3 KB of scratch per lane
  if constexpr (M * P > 512 || lds > 144 * 1024) {
    return false;
  } else {
    if (probe_only) return true;
    a.tiles_per_outer = (a.ncols + 7) / 8;
    static PerDeviceOnce configured;
    if (auto once = configured.first_use()) {
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_xfused_mixed<M, P>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL((k_xfused_mixed<M, P>), dim3((unsigned)((long)a.tiles_per_outer * nouter)), dim3(M * P), lds, s, a, wN);
    FG_HIP_CHECK(hipGetLastError());
    return true;
  }
}

template <int P>
bool xfused_mixed_p(int m, const XFusedArgs& a, int nouter, const cplx* wN, hipStream_t s, bool probe_only) {
  switch (m) {
    case 8: return launch_xfused_mixed<8, P>(a, nouter, wN, s, probe_only);
    case 16: return launch_xfused_mixed<16, P>(a, nouter, wN, s, probe_only);
    case 32: return launch_xfused_mixed<32, P>(a, nouter, wN, s, probe_only);
    case 64: return launch_xfused_mixed<64, P>(a, nouter, wN, s, probe_only);
    case 128: return launch_xfused_mixed<128, P>(a, nouter, wN, s, probe_only);
    case 256: return launch_xfused_mixed<256, P>(a, nouter, wN, s, probe_only);
    default: return false;
  }
}

template <int P>
bool strided_mixed_p(int m, const StridedArgs& a, int nouter, int dir, int ncomp, long cs, const cplx* wN, hipStream_t s) {
  switch (m) {
    case 8: return launch_strided_mixed<8, P>(a, nouter, dir, ncomp, cs, wN, s);
    case 16: return launch_strided_mixed<16, P>(a, nouter, dir, ncomp, cs, wN, s);
    case 32: return launch_strided_mixed<32, P>(a, nouter, dir, ncomp, cs, wN, s);
    case 64: return launch_strided_mixed<64, P>(a, nouter, dir, ncomp, cs, wN, s);
    case 128: return launch_strided_mixed<128, P>(a, nouter, dir, ncomp, cs, wN, s);
    case 256: return launch_strided_mixed<256, P>(a, nouter, dir, ncomp, cs, wN, s);
    default: return false;
  }
}

// z axis, nz = 2 M, M = p M': after the M'-point sub-transforms of the packed rows (Y_r[k'] at complex p k' + r), one
// sweep per row forms the M-point spectrum Z and splits it into the half spectrum of the real row (r2c_split):
// thread k <= M/2 writes X[k] and X[M - k].  wn: e^{-2 pi i j / nz}, j < nz  (w_M^j = wn[2 j]).
// Both z sweeps work in place: a workgroup owns `rows` whole rows, forms their outputs in an LDS image and writes them
// back after a barrier (no scratch component, no copy).
template <int P>
__global__ __launch_bounds__(256) void k_mixed_r2c_finish(double* data, long nrows, int nzp, int M, int p_rt, int rows,
                                                          const cplx* wn) {
  extern __shared__ __align__(16) double lds_raw[];
  cplx* img = reinterpret_cast<cplx*>(lds_raw);   // [rows][M + 1]
  const int p = P ? P : p_rt;
  const int half = M / 2 + 1;
  const long row0 = (long)blockIdx.x * rows;
  const int Mp = M / p, nz = 2 * M;
  for (int it = threadIdx.x; it < rows * half; it += 256) {
  const int lr = it / half;
  const long row = row0 + lr;
  if (row >= nrows) break;
  const int k = it % half;
  const cplx* in = reinterpret_cast<const cplx*>(data + row * nzp);
  cplx* out = img + (long)lr * (M + 1);
  auto zfull = [&](int K) {   // Z[K], K in [0, M)
    const int kk = K % Mp, sidx = K / Mp;
    cplx acc = cmake(0.0, 0.0);
    const int step = 2 * (kk + sidx * Mp);   // = 2 K < nz
    int idx = 0;
#pragma unroll
    for (int r = 0; r < p; ++r) {
      acc = cadd(acc, cmul(in[p * kk + r], wn[idx]));
      idx += step;
      if (idx >= nz) idx -= nz;
    }
    return acc;
  };
  const cplx zk = zfull(k), zm = zfull((M - k) % M);
  out[k] = r2c_split(zk, zm, wn[k]);
  out[M - k] = r2c_split(zm, zk, wn[M - k]);   // k = 0 writes the Nyquist bin X[M] = split(Z[0], Z[0])
  }
  __syncthreads();
  for (int it = threadIdx.x; it < rows * (M + 1); it += 256) {
    const int lr = it / (M + 1);
    if (row0 + lr >= nrows) break;
    reinterpret_cast<cplx*>(data + (row0 + lr) * nzp)[it % (M + 1)] = img[it];
  }
}

// inverse: merge the half spectrum into Z' (c2r_merge), then the inverse combine to the sub-rows:
// thread k' < M' reads X[k' + M' s], X[M - k' - M' s] and writes Z_r[k'] to complex p k' + r.
template <int P>
__global__ __launch_bounds__(256) void k_mixed_c2r_start(double* data, long nrows, int nzp, int M, int p_rt, int rows,
                                                         const cplx* wn) {
  extern __shared__ __align__(16) double lds_raw[];
  cplx* img = reinterpret_cast<cplx*>(lds_raw);   // [rows][M]
  const int p = P ? P : p_rt;
  const int Mp = M / p, nz = 2 * M;
  const long row0 = (long)blockIdx.x * rows;
  for (int it = threadIdx.x; it < rows * Mp; it += 256) {
  const int lr = it / Mp;
  const long row = row0 + lr;
  if (row >= nrows) break;
  const int kk = it % Mp;
  const cplx* in = reinterpret_cast<const cplx*>(data + row * nzp);
  cplx* out = img + (long)lr * M;
  cplx z[P ? P : kMaxOddFactor];
#pragma unroll
  for (int sidx = 0; sidx < p; ++sidx) {
    const int K = kk + Mp * sidx;
    cplx xk = in[K], xm = in[M - K];
    if (K == 0) { xk.im = 0.0; xm.im = 0.0; }   // FFTW's c2r ignores the imaginary parts of the DC and Nyquist bins
    z[sidx] = c2r_merge(xk, xm, wn[K]);
  }
#pragma unroll
  for (int r = 0; r < p; ++r) {
    cplx acc = cmake(0.0, 0.0);
    const int step = (int)((2L * r * Mp) % nz);
    int idx = 0;
#pragma unroll
    for (int sidx = 0; sidx < p; ++sidx) {
      acc = cadd(acc, cmul(z[sidx], cconj(wn[idx])));
      idx += step;
      if (idx >= nz) idx -= nz;
    }
    out[p * kk + r] = cmul(acc, cconj(wn[2 * r * kk]));   // 2 r k' < nz
  }
  }
  __syncthreads();
  for (int it = threadIdx.x; it < rows * M; it += 256) {
    const int lr = it / M;
    if (row0 + lr >= nrows) break;
    reinterpret_cast<cplx*>(data + (row0 + lr) * nzp)[it % M] = img[it];
  }
}
#pragma clang diagnostic pop

cplx* upload(const std::vector<cplx>& v) {
  cplx* d = nullptr;
  FG_HIP_CHECK(hipMalloc(&d, v.size() * sizeof(cplx)));
  FG_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(cplx), hipMemcpyHostToDevice));
  return d;
}

template <class K>
void launch_strided(const StridedArgs& a, long nblocks, int ncomp, long comp_stride, hipStream_t s) {
  static PerDeviceOnce configured;
  const size_t lds = K::LDS_DOUBLES * sizeof(double);
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_strided<K>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(k_strided<K>, dim3((unsigned)nblocks, ncomp), dim3(K::THREADS), lds, s, a, comp_stride);
  FG_HIP_CHECK(hipGetLastError());
}

template <class K>
void launch_plane(const PlaneArgs& a, int nplanes, int ncomp, long comp_stride, hipStream_t s) {
  static PerDeviceOnce configured;
  const size_t lds = K::LDS_DOUBLES * sizeof(double);
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_plane<K>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(k_plane<K>, dim3((unsigned)nplanes, ncomp), dim3(K::THREADS), lds, s, a, comp_stride);
  FG_HIP_CHECK(hipGetLastError());
}

// (ny, nz / 2) pairs the plane kernels are built for: THREADS = ny * nz / 16 <= 1024, the complex plane within the LDS
template <int NY, int M>
bool plane_nm(int ny, int m, bool probe, const PlaneArgs& a, int nplanes, int ncomp, long cs, int dir, hipStream_t s) {
  if (ny != NY || m != M) return false;
  if (probe) return true;
  if (dir < 0) launch_plane<ZYKernel<NY, M>>(a, nplanes, ncomp, cs, s);
  else launch_plane<YZKernel<NY, M>>(a, nplanes, ncomp, cs, s);
  return true;
}
bool plane_dispatch(int ny, int m, bool probe, const PlaneArgs& a, int nplanes, int ncomp, long cs, int dir, hipStream_t s) {
#define FG_PLANE(NY, M) plane_nm<NY, M>(ny, m, probe, a, nplanes, ncomp, cs, dir, s)
  return FG_PLANE(16, 8) || FG_PLANE(16, 16) || FG_PLANE(16, 32) || FG_PLANE(16, 64) || FG_PLANE(32, 8) || FG_PLANE(32, 16) ||
         FG_PLANE(32, 32) || FG_PLANE(32, 64) || FG_PLANE(64, 8) || FG_PLANE(64, 16) || FG_PLANE(64, 32) || FG_PLANE(64, 64) ||
         FG_PLANE(128, 8) || FG_PLANE(128, 16) || FG_PLANE(128, 32) || FG_PLANE(128, 64) || FG_PLANE(256, 8) ||
         FG_PLANE(256, 16) || FG_PLANE(256, 32);   // (256, 64): 264 KB
#undef FG_PLANE
}

template <class K>
void launch_z(const ZArgs& a, int ncomp, long comp_stride, int lines, hipStream_t s) {
  static PerDeviceOnce configured;
  const size_t lds = K::LDS_DOUBLES * sizeof(double);
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_zpass<K>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  long nblocks = (a.nrows + lines - 1) / lines;
  hipLaunchKernelGGL(k_zpass<K>, dim3((unsigned)nblocks, ncomp), dim3(K::THREADS), lds, s, a, comp_stride);
  FG_HIP_CHECK(hipGetLastError());
}

template <int N, int C, int NC, bool XSPLIT = false, int NTC = -1>
void xfused_nc(XFusedArgs a, int nouter, hipStream_t s);

template <int N>
void xfused_n(XFusedArgs a, int nouter, int ncomp, hipStream_t s) {
  // (half-segment tiles, C = 4, were measured for N = 512: 2.96 ms against 2.5 ms)
  // three components: 8-column (128-byte) tiles whatever the length -- more, smaller workgroups balance better over the
  // 256 CUs than 256-thread tiles of short lines (128^3: 0.037 -> 0.035 ms, 6 220 -> 6 315 it/s; 64^3: +1 %)
  if constexpr (N == 1024) {
    // 1024-point lines: the exchange buffer of an 8-column tile (147 KB) leaves no room for 512 threads' worth of registers
    // (16 waves: 128 VGPRs each, the three spectra alone take 96) -- half-segment tiles, still one pass instead of three
    if (ncomp == 1) xfused_nc<N, 4, 1>(a, nouter, s);   // (8 columns = 16 waves at 128 VGPRs: 33 spilled)
    else if (a.xjump != 0) throw std::runtime_error("fft: interleaved slab layout: x lines up to 512");
    else xfused_nc<N, 4, 3>(a, nouter, s);
  } else {
    // (r5: the scalar modes' pass keeps its pass twiddles in LDS like the three-component one -- XFusedKernel::TW_LDS -- 256^3 porous
    // K4 0.083 -> 0.073 ms, 2 765 -> 2 830 it/s; on cache-resident grids also the branch-free loads, 128^3 porous 14 030 -> 14 370)
    if (ncomp == 1 && (N == 64 || N == 128) && a.nt == 0)
      xfused_nc<N, XTileCols<N>::value, 1, false, ((N == 64 || N == 128) ? 16 : -1)>(a, nouter, s);
    else if (ncomp == 1) xfused_nc<N, XTileCols<N>::value, 1>(a, nouter, s);
    // Cache-resident fields (nt = 0: three components within the Infinity Cache -- 128^3 and below, and the slabs of 256^3 on 8
    // GPUs) with lines up to 256 points: the tile's loads without branches and all issued up front, so that the first component's
    // transform starts when ITS eight loads have landed (s_waitcnt vmcnt(16)) instead of after all 24.  r5, one job each: K4 at
    // 128^3 31.5 -> 28.5 us (7 560 -> 7 705 it/s), 64^3 10.0 -> 9.3 us; one slab of eight at 256^3 (interleaved layout) 42.6 ->
    // 38.9 us, 256 x 32 x 256 33.4 -> 30.0 us; 256^3 and 512^3 on one GPU (streaming, nt = 3) measured equal and keep the
    // run-time flags
    else if (a.xjump != 0 && (N == 64 || N == 128 || N == 256) && a.nt == 0)
      xfused_nc<N, 8, 3, true, ((N == 64 || N == 128 || N == 256) ? 16 : -1)>(a, nouter, s);
    else if (a.xjump != 0) xfused_nc<N, 8, 3, true>(a, nouter, s);   // slab decomposition, components interleaved per peer
    else if ((N == 64 || N == 128 || N == 256) && a.nt == 0)
      xfused_nc<N, 8, 3, false, ((N == 64 || N == 128 || N == 256) ? 16 : -1)>(a, nouter, s);
    else xfused_nc<N, 8, 3>(a, nouter, s);
  }
}

template <int N, int C, int NC, bool XSPLIT, int NTC>
void xfused_nc(XFusedArgs a, int nouter, hipStream_t s) {
  using K = XFusedKernel<N, C, NC, XSPLIT, NTC>;
  static PerDeviceOnce configured;
  const size_t lds = K::LDS_DOUBLES * sizeof(double);
  if (auto once = configured.first_use()) {
    FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_xfused<K>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  a.tiles_per_outer = (a.ncols + C - 1) / C;
  const long nblocks = (long)a.tiles_per_outer * nouter;
  a.xcd_order = 1;   // every XCD a contiguous run of tiles (512^3 1.84 -> 1.80 ms, 256^3 0.200 -> 0.188 ms)
  static cplx xq[8];
  static bool have_xq = false;
  if (!have_xq) {
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int q = 0; q < 8; ++q) {
      const long double th = pi * (long double)fft::Line<N>::last_index(0, q) / (long double)N;
      xq[q] = cmake((double)cosl(th), (double)sinl(th));
    }
    have_xq = true;
  }
  for (int q = 0; q < 8; ++q) a.xq[q] = xq[q];
  // (persistent / looping forms of this kernel lose to fresh workgroups: DESIGN 3.4, 3.6)
  hipLaunchKernelGGL(k_xfused<K>, dim3((unsigned)nblocks), dim3(K::THREADS), lds, s, a);
  FG_HIP_CHECK(hipGetLastError());
}

template <int N>
void strided_n(const StridedArgs& a0, int nouter, int dir, int ncomp, long comp_stride, hipStream_t s) {
  constexpr int C = TileCols<N>::value;
  StridedArgs a = a0;
  a.tiles_per_outer = (a.ncols + C - 1) / C;
  long nblocks = (long)a.tiles_per_outer * nouter;
  if (dir < 0) launch_strided<StridedKernel<N, C, -1>>(a, nblocks, ncomp, comp_stride, s);
  else launch_strided<StridedKernel<N, C, +1>>(a, nblocks, ncomp, comp_stride, s);
}

// the same with 8-column tiles whatever the length (sub-rows of the z pass have only p <= 15 columns)
template <int N>
void strided_n8(const StridedArgs& a0, int nouter, int dir, int ncomp, long comp_stride, hipStream_t s) {
  StridedArgs a = a0;
  a.tiles_per_outer = (a.ncols + 7) / 8;
  long nblocks = (long)a.tiles_per_outer * nouter;
  if (dir < 0) launch_strided<StridedKernel<N, 8, -1>>(a, nblocks, ncomp, comp_stride, s);
  else launch_strided<StridedKernel<N, 8, +1>>(a, nblocks, ncomp, comp_stride, s);
}
void strided_pow2_narrow(int n, const StridedArgs& a, int nouter, int dir, int ncomp, long cs, hipStream_t s) {
  switch (n) {
    case 8: strided_n8<8>(a, nouter, dir, ncomp, cs, s); break;
    case 16: strided_n8<16>(a, nouter, dir, ncomp, cs, s); break;
    case 32: strided_n8<32>(a, nouter, dir, ncomp, cs, s); break;
    case 64: strided_n8<64>(a, nouter, dir, ncomp, cs, s); break;
    case 128: strided_n8<128>(a, nouter, dir, ncomp, cs, s); break;
    case 256: strided_n8<256>(a, nouter, dir, ncomp, cs, s); break;
    case 512: strided_n8<512>(a, nouter, dir, ncomp, cs, s); break;
    case 1024: strided_n8<1024>(a, nouter, dir, ncomp, cs, s); break;
    default: throw std::runtime_error("fft: unsupported fast length");
  }
}

void strided_pow2(int n, const StridedArgs& a, int nouter, int dir, int ncomp, long cs, hipStream_t s) {
  // short lines use 16- to 256-column tiles (256 threads); when the columns do not fill the last tile of a row -- 128^3:
  // 72 columns = 4.5 tiles of 16 -- 8-column tiles waste nothing and balance better: y passes 0.028 -> 0.026 ms,
  // 128^3 6 365 -> 6 590 it/s, 64^3 +2 %
  if (n <= 128 && a.ncols % (2048 / n > 8 ? 2048 / n : 8) != 0) return strided_pow2_narrow(n, a, nouter, dir, ncomp, cs, s);
  switch (n) {
    case 8: strided_n<8>(a, nouter, dir, ncomp, cs, s); break;
    case 16: strided_n<16>(a, nouter, dir, ncomp, cs, s); break;
    case 32: strided_n<32>(a, nouter, dir, ncomp, cs, s); break;
    case 64: strided_n<64>(a, nouter, dir, ncomp, cs, s); break;
    case 128: strided_n<128>(a, nouter, dir, ncomp, cs, s); break;
    case 256: strided_n<256>(a, nouter, dir, ncomp, cs, s); break;
    case 512: strided_n<512>(a, nouter, dir, ncomp, cs, s); break;
    case 1024: strided_n<1024>(a, nouter, dir, ncomp, cs, s); break;
    default: throw std::runtime_error("fft: unsupported fast length");
  }
}

}  // namespace

Fft3::Fft3(const Grid& g, hipStream_t stream) : g_(g), stream_(stream), wz_(nullptr), scratch_(nullptr) {
  const int len[3] = {g.nx, g.ny, g.nz};
  for (int a = 0; a < 3; ++a) {
    tw_[a] = nullptr;
    wgen_[a] = nullptr;
  }
  half_root_[0] = half_root_[1] = nullptr;
  fast_[0] = fast_len(g.nx);
  fast_[1] = fast_len(g.ny);
  fast_[2] = (g.nz % 2 == 0) && fast_len(g.nz / 2);
  bool need_scratch = false;
  for (int a = 0; a < 3; ++a) {
    odd_[a] = 0;
    if (fast_[a]) {
      tw_[a] = upload(make_pass_twiddles(a == 2 ? len[a] / 2 : len[a]));
      if (a < 2) half_root_[a] = upload(make_unit_roots(2 * len[a], len[a] / 8));
    } else {
      // p * 2^k: power-of-two kernels on the interleaved sub-lines + a combine sweep; anything else: O(n^2) DFT
      const int m = a == 2 ? (len[a] % 2 == 0 ? len[a] / 2 : 0) : len[a];
      if (a == 2 && len[a] % 2 == 1 && len[a] > 1) {   // odd nz: no packed-real trick, the rows as nz complex points
        SmoothPlan so;
        if (smooth_plan_z(len[a], &so)) smooth_[2] = so, zodd_ = true;
      }
      odd_[a] = m ? mixed_factor(m) : 0;
      if (odd_[a]) tw_[a] = upload(make_pass_twiddles(m / odd_[a]));
      // lengths with small prime factors that the one-kernel p * 2^k passes (p = 3, 5, 7, 9) do not cover -- 100, 120, 200, 300,
      // 400, 500 ...: the Stockham tile kernels of fg_fft_smooth.h.  (Where a p * 2^k kernel exists it stays: measured in one job
      // with the tile kernels forced on, 96^3 9 160 against 5 510 it/s, 192^3 2 030 / 1 270, 384^3 240 / 174, 448^3 123 / 91.)
      SmoothPlan sp;
      // ... and, since the tile kernels are built per plan (fg_fft_smooth_plans.h), from 100 points on (z: nz / 2 >= 72): sub-line
      // kernels / tile kernels in one job 144^3 3 034 / 4 452 it/s, 160^3 3 119 / 3 647, 192^3 2 050 / 2 190, 224^3 965 / 1 237,
      // 288^3 432 / 657, 320^3 409 / 481, 448^3 121 / 148; 96^3 9 281 / 9 295 (kept on the sub-line kernels)
      const bool one_kernel_mixed = (odd_[a] == 3 || odd_[a] == 5 || odd_[a] == 7 || odd_[a] == 9) && m <= 1024 && m < (a == 2 ? 72 : 100);
      if (m > 1 && !one_kernel_mixed && (a == 2 ? smooth_plan_z(m, &sp) : smooth_plan_strided(m, &sp))) smooth_[a] = sp;
      wgen_[a] = upload(make_unit_roots(len[a], len[a]));
      need_scratch = true;
    }
  }
  if (smooth_[0].n) {   // plans of the tile kernels' fused x pass (one / three components)
    smooth_plan_xfused(g.nx, 1, &xfused_plan_[0]);
    smooth_plan_xfused(g.nx, 3, &xfused_plan_[1]);
    // an x length whose three components have no joint image plan (384 = 3 * 2^7: 9 216 points of a tile against 512 threads x 20
    // values) keeps the sub-line kernels' fused x pass where there is one: 384^3 fused x 1 114 us against 1 732 us in the
    // R <= 32 class kernel, y and z on the tile kernels either way (238 -> 263 it/s)
    if (xfused_plan_[1].joint != 3 && (odd_[0] == 3 || odd_[0] == 5)) {
      XFusedArgs probe = {};
      const bool mixed_fused = odd_[0] == 3 ? xfused_mixed_p<3>(g.nx / 3, probe, 0, nullptr, stream_, true)
                                            : xfused_mixed_p<5>(g.nx / 5, probe, 0, nullptr, stream_, true);
      if (mixed_fused) smooth_[0] = xfused_plan_[0] = xfused_plan_[1] = SmoothPlan();
    }
  }
  if (fast_[2]) wz_ = upload(make_unit_roots(g.nz, g.nz / 2 + 1));
  if (need_scratch) FG_HIP_CHECK(hipMalloc(&scratch_, g.n * sizeof(double)));
  // three components larger than the 256 MB Infinity Cache: nothing a pass writes is still cached when the next reads it
  stream_stores_ = 3.0 * (double)g.n * sizeof(double) > 256.0 * 1024 * 1024 ? 1 : 0;
}

void Fft3::set_joint_x(bool on) {
  if (on == joint_x_) return;
  joint_x_ = on;
  if (smooth_[0].n) {
    smooth_plan_xfused(g_.nx, 1, &xfused_plan_[0], on);
    smooth_plan_xfused(g_.nx, 3, &xfused_plan_[1], on);
  }
}

Fft3::~Fft3() {
  for (int a = 0; a < 3; ++a) {
    if (tw_[a]) (void)hipFree(tw_[a]);
    if (a < 2 && half_root_[a]) (void)hipFree(half_root_[a]);
    if (wgen_[a]) (void)hipFree(wgen_[a]);
  }
  if (wz_) (void)hipFree(wz_);
  if (scratch_) (void)hipFree(scratch_);
}

void Fft3::strided(double* data, int ncomp, long comp_stride, int axis, int dir, double scale, const PlaneWindow* w) {
  const int n = axis == 0 ? g_.nx : g_.ny;
  const long ls = axis == 0 ? (long)g_.ny * g_.nzc : g_.nzc;
  const int ncols = axis == 0 ? g_.ny * g_.nzc : g_.nzc;
  int nouter = axis == 0 ? 1 : g_.nx;
  const long os = axis == 0 ? 0 : (long)g_.ny * g_.nzc;
  const bool windowed = w && w->np >= 0;
  if (windowed && (axis != 1 || !fast_[1] || w->x0 < 0 || w->x0 + w->np > g_.nx))
    throw std::runtime_error("fft: plane windows need a power-of-two y pass");
  if (n == 1) return;  // identity; forward() never folds the scale into a length-1 axis
  if (fast_[axis]) {
    StridedArgs a;
    a.data = reinterpret_cast<cplx*>(data);
    if (windowed) a.data += (long)w->x0 * os, nouter = w->np;
    if (nouter == 0) return;
    a.ls = ls;
    a.os = os;
    a.ncols = ncols;
    a.tiles_per_outer = 0;
    a.scale = scale;
    a.tw = tw_[axis];
    a.nt = stream_stores_ ? (1 | ((nt_loads_env() & 1) ? 2 : 0)) : 0;
    if (w && w->nt >= 0) a.nt = w->nt;
    a.xcd_order = 0;   // (measured for the y passes: 512^3 -1..2 %, 256^3 +2 %)
    strided_pow2(n, a, nouter, dir, ncomp, comp_stride / 2, stream_);
    return;
  }
  if (smooth_[axis].n) {
    SmoothArgs a;
    a.data = reinterpret_cast<cplx*>(data);
    a.ls = ls;
    a.os = os;
    a.ncols = ncols;
    a.tiles_per_outer = 0;
    a.scale = scale;
    a.w = wgen_[axis];
    a.nt = stream_stores_ ? 3 : 0;
    a.plan = smooth_[axis];
    launch_smooth_strided(a, nouter, dir, ncomp, comp_stride / 2, stream_);
    return;
  }
  if (odd_[axis]) {
    // n = p * m: m-point kernels on the p interleaved sub-lines, combine sweep through the scratch component
    const int p = odd_[axis], m = n / p;
    if (p == 3 || p == 5 || p == 7 || p == 9) {
      StridedArgs a;
      a.data = reinterpret_cast<cplx*>(data);
      a.ls = ls;
      a.os = os;
      a.ncols = ncols;
      a.tiles_per_outer = 0;
      a.scale = scale;
      a.tw = tw_[axis];
      a.nt = stream_stores_ ? 3 : 0;
      a.xcd_order = 0;
      const long cs2 = comp_stride / 2;
      const bool done = p == 3   ? strided_mixed_p<3>(m, a, nouter, dir, ncomp, cs2, wgen_[axis], stream_)
                        : p == 5 ? strided_mixed_p<5>(m, a, nouter, dir, ncomp, cs2, wgen_[axis], stream_)
                        : p == 7 ? strided_mixed_p<7>(m, a, nouter, dir, ncomp, cs2, wgen_[axis], stream_)
                                 : strided_mixed_p<9>(m, a, nouter, dir, ncomp, cs2, wgen_[axis], stream_);
      if (done) return;
    }
    const long total = (long)nouter * m * ncols;
    const unsigned nb = (unsigned)((total + 255) / 256);
    auto subs = [&]() {
      for (int r = 0; r < p; ++r) {
        StridedArgs a;
        a.data = reinterpret_cast<cplx*>(data) + r * ls;
        a.ls = ls * p;
        a.os = os;
        a.ncols = ncols;
        a.tiles_per_outer = 0;
        a.scale = 1.0;
        a.tw = tw_[axis];
        a.nt = 0;
        a.xcd_order = 0;
        strided_pow2(m, a, nouter, dir, ncomp, comp_stride / 2, stream_);
      }
    };
    const size_t tile_lds = (size_t)n * 8 * sizeof(cplx);
    auto combine_tile = [&]() {
      const int tiles = (ncols + 7) / 8;
      const dim3 grid((unsigned)((long)tiles * nouter));
      for (int c = 0; c < ncomp; ++c) {
        cplx* d = reinterpret_cast<cplx*>(data + c * comp_stride);
#define FG_TILE_COMBINE(D, PP)                                                                                         \
  do {                                                                                                                 \
    static PerDeviceOnce configured;                                                                                    \
    if (auto once = configured.first_use()) {                                                                                                 \
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mixed_combine_tile<D, PP>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                       \
    }                                                                                                                  \
    hipLaunchKernelGGL((k_mixed_combine_tile<D, PP>), grid, dim3(256), tile_lds, stream_, d, ls, os, ncols, tiles, m, p, \
                       scale, wgen_[axis]);                                                                            \
  } while (0)
        if (dir < 0) {
          if (p == 3) FG_TILE_COMBINE(-1, 3);
          else if (p == 5) FG_TILE_COMBINE(-1, 5);
          else if (p == 7) FG_TILE_COMBINE(-1, 7);
          else if (p == 9) FG_TILE_COMBINE(-1, 9);
          else if (p == 15) FG_TILE_COMBINE(-1, 15);
          else if (p == 25) FG_TILE_COMBINE(-1, 25);
          else FG_TILE_COMBINE(-1, 0);
        } else {
          if (p == 3) FG_TILE_COMBINE(+1, 3);
          else if (p == 5) FG_TILE_COMBINE(+1, 5);
          else if (p == 7) FG_TILE_COMBINE(+1, 7);
          else if (p == 9) FG_TILE_COMBINE(+1, 9);
          else if (p == 15) FG_TILE_COMBINE(+1, 15);
          else if (p == 25) FG_TILE_COMBINE(+1, 25);
          else FG_TILE_COMBINE(+1, 0);
        }
#undef FG_TILE_COMBINE
        FG_HIP_CHECK(hipGetLastError());
      }
    };
    auto combine_scratch = [&]() {
      for (int c = 0; c < ncomp; ++c) {
        cplx* src = reinterpret_cast<cplx*>(data + c * comp_stride);
#define FG_COMBINE(D, PP)                                                                                              \
  hipLaunchKernelGGL((k_mixed_combine<D, PP>), dim3(nb), dim3(256), 0, stream_, src, reinterpret_cast<cplx*>(scratch_), ls, \
                     os, ncols, nouter, m, p, scale, wgen_[axis])
        if (dir < 0) {
          if (p == 3) FG_COMBINE(-1, 3);
          else if (p == 5) FG_COMBINE(-1, 5);
          else if (p == 7) FG_COMBINE(-1, 7);
          else FG_COMBINE(-1, 0);
        } else {
          if (p == 3) FG_COMBINE(+1, 3);
          else if (p == 5) FG_COMBINE(+1, 5);
          else if (p == 7) FG_COMBINE(+1, 7);
          else FG_COMBINE(+1, 0);
        }
#undef FG_COMBINE
        FG_HIP_CHECK(hipGetLastError());
        FG_HIP_CHECK(hipMemcpyAsync(src, scratch_, g_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
      }
    };
    auto combine = [&]() {
      if (tile_lds <= 144 * 1024) combine_tile();   // whole lines in LDS: in place
      else combine_scratch();
    };
    if (dir < 0) {
      subs();
      combine();
    } else {
      combine();
      subs();
    }
    return;
  }
  // generic: component by component through the scratch buffer
  const long total = (long)nouter * n * ncols;
  const int bs = 256;
  for (int c = 0; c < ncomp; ++c) {
    cplx* src = reinterpret_cast<cplx*>(data + c * comp_stride);
    hipLaunchKernelGGL(k_dft_strided_generic, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, stream_, src,
                       reinterpret_cast<cplx*>(scratch_), ls, os, ncols, nouter, n, dir, scale, wgen_[axis]);
    FG_HIP_CHECK(hipGetLastError());
    FG_HIP_CHECK(hipMemcpyAsync(src, scratch_, g_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  }
}

// forward transform along `axis` (0 = x of [nx][ny][nzc], 1 = y, used for the x lines of a y-slab),
// scale, Green operator, inverse transform -- one kernel for the three components.
bool Fft3::can_fuse(int axis, int ncomp) const {
  const int n = axis == 0 ? g_.nx : g_.ny;
  if (fast_[axis]) return n <= 1024;
  if (axis == 0 && smooth_[0].n && (ncomp == 1 || ncomp == 3))   // the tile kernels' fused x pass: ncomp images in LDS
    return xfused_plan_[ncomp == 3].n != 0;
  // p * 2^k with p = 3, 5: the three-component form only
  if (ncomp != 3 || (odd_[axis] != 3 && odd_[axis] != 5)) return false;
  XFusedArgs a = {};
  return odd_[axis] == 3 ? xfused_mixed_p<3>(n / 3, a, 0, nullptr, stream_, true) : xfused_mixed_p<5>(n / 5, a, 0, nullptr, stream_, true);
}

bool Fft3::can_plane() const {
  if (!fast_[1] || !fast_[2] || g_.nz % 2) return false;
  return plane_dispatch(g_.ny, g_.nz / 2, true, PlaneArgs{}, 0, 0, 0, 0, stream_);
}

void Fft3::zy_plane(double* data, int ncomp, long comp_stride, int dir) {
  PlaneArgs a = {data, (long)g_.ny * g_.nzp, g_.nzp, tw_[2], wz_, tw_[1]};
  if (!can_plane() || !plane_dispatch(g_.ny, g_.nz / 2, false, a, g_.nx, ncomp, comp_stride, dir, stream_))
    throw std::runtime_error("fft: plane kernels not available for this grid");
}

bool Fft3::can_xlayout() const {
  return fast_[0] && fast_[1] && g_.nx >= 8 && g_.nx <= 512 && g_.nzc % 8 == 0 && g_.ny >= 8;
}

void Fft3::c2c_y_xlayout(double* in, long in_cs, double* out, long out_cs, int ncomp, int dir, double scale, const PlaneWindow* w) {
  if (!can_xlayout()) throw std::runtime_error("fft: x-contiguous layout not available for this grid");
  const bool windowed = w && w->np >= 0;
  if (windowed && (w->x0 < 0 || w->x0 + w->np > g_.nx)) throw std::runtime_error("fft: plane window out of range");
  const int x0 = windowed ? w->x0 : 0, np = windowed ? w->np : g_.nx;
  if (np == 0) return;
  StridedArgs a;
  a.data = reinterpret_cast<cplx*>(in);
  a.out = reinterpret_cast<cplx*>(out);
  a.out_cs = out_cs / 2;
  a.ncols = g_.nzc;
  a.tiles_per_outer = 0;
  a.scale = scale;
  a.tw = tw_[1];
  a.nt = stream_stores_ ? (1 | ((nt_loads_env() & 1) ? 2 : 0)) : 0;
  if (w && w->nt >= 0) a.nt = w->nt;
  a.xcd_order = 0;
  const long plain_ls = g_.nzc, plain_os = (long)g_.ny * g_.nzc;
  const long xl_ls = (long)g_.nx * 8, xl_os = 8, xl_ts = (long)g_.ny * g_.nx * 8;
  if (dir < 0) {
    a.ls = plain_ls;
    a.os = plain_os;
    a.ls_out = xl_ls;
    a.os_out = xl_os;
    a.ts_out = xl_ts;
    a.data += (long)x0 * plain_os;
    a.out += (long)x0 * xl_os;
  } else {
    a.ls = xl_ls;
    a.os = xl_os;
    a.ts_in = xl_ts;
    a.ls_out = plain_ls;
    a.os_out = plain_os;
    a.data += (long)x0 * xl_os;
    a.out += (long)x0 * plain_os;
  }
  strided_pow2_narrow(g_.ny, a, np, dir, ncomp, in_cs / 2, stream_);   // 8-column tiles: the layout's inner dimension
}

void Fft3::fused_g0(double* data, long comp_stride, int axis, double scale, const G0Params& gp, int jj0, int ncomp, int xsplit,
                    long xjump, bool xlayout) {
  if (ncomp != 1 && ncomp != 3) throw std::runtime_error("fft: fused Green-operator pass takes 1 or 3 components");
  if (!can_fuse(axis, ncomp)) throw std::runtime_error("fft: fused Green-operator pass not available for this length");
  const int n = axis == 0 ? g_.nx : g_.ny;
  XFusedArgs a;
  a.data = reinterpret_cast<cplx*>(data);
  a.comp_stride = comp_stride / 2;
  a.ls = axis == 0 ? (long)g_.ny * g_.nzc : g_.nzc;
  a.os = axis == 0 ? 0 : (long)g_.ny * g_.nzc;
  a.ncols = axis == 0 ? g_.ny * g_.nzc : g_.nzc;
  a.tiles_per_outer = 0;
  a.flat_cols = axis == 0 ? 1 : 0;
  a.nzc = g_.nzc;
  a.nzf = g_.nzf;
  a.jj0 = jj0;
  a.scale = scale;
  a.c10 = gp.c10;
  a.c20 = gp.c20;
  a.tw = tw_[axis];
  a.half_root = half_root_[axis];
  a.inv_h = gp.inv_h0;
  a.nt = stream_stores_ ? (1 | ((nt_loads_env() & 2) ? 2 : 0)) : 0;
  for (int k = 0; k < 3; ++k) {
    a.kpm[k] = gp.kpm[k];
    a.kp[k] = gp.kp[k];
  }
  if (xjump != 0) {
    if (!fast_[axis] || axis != 0 || ncomp != 3) throw std::runtime_error("fft: interleaved layout needs the radix x pass");
    a.xsplit = xsplit;
    a.xjump = xjump / 2;
  }
  if (xlayout) {
    if (!can_xlayout() || axis != 0 || ncomp != 3 || xjump != 0) throw std::runtime_error("fft: x-contiguous layout: three components, x pass");
    a.xl_ny = g_.ny;
    a.ls = 8;
    a.flat_cols = 0;
    a.ncols = g_.ny * g_.nzc;   // (tiles = ncols / 8 = (nzc / 8) * ny)
  }
  const int nouter = axis == 0 ? 1 : g_.nx;
  if (!fast_[axis] && axis == 0 && smooth_[0].n) {
    if (xjump != 0 || xlayout) throw std::runtime_error("fft: the tile kernels' fused x pass takes the plain layout");
    SmoothXArgs x;
    x.base.plan = xfused_plan_[ncomp == 3];
    if (!x.base.plan.n) throw std::runtime_error("fft: fused Green-operator pass not available for this length");
    x.base.data = reinterpret_cast<cplx*>(data);
    x.base.ls = (long)g_.ny * g_.nzc;
    x.base.os = 0;
    x.base.ncols = g_.ny * g_.nzc;
    x.base.tiles_per_outer = 0;
    x.base.scale = scale;
    x.base.w = wgen_[0];
    x.base.nt = stream_stores_ ? 3 : 0;
    x.comp_stride = comp_stride / 2;
    x.ncomp = ncomp;
    x.nzc = g_.nzc;
    x.nzf = g_.nzf;
    x.jj0 = jj0;
    x.c10 = gp.c10;
    x.c20 = gp.c20;
    for (int k = 0; k < 3; ++k) x.kpm[k] = gp.kpm[k], x.kp[k] = gp.kp[k];
    launch_smooth_xfused(x, stream_);
    return;
  }
  if (!fast_[axis]) {
    a.nt = stream_stores_ ? 3 : 0;
    const bool ok = odd_[axis] == 3 ? xfused_mixed_p<3>(n / 3, a, nouter, wgen_[axis], stream_, false)
                                    : xfused_mixed_p<5>(n / 5, a, nouter, wgen_[axis], stream_, false);
    if (!ok) throw std::runtime_error("fft: fused Green-operator pass not available for this length");
    return;
  }
  switch (n) {
    case 8: xfused_n<8>(a, nouter, ncomp, stream_); break;
    case 16: xfused_n<16>(a, nouter, ncomp, stream_); break;
    case 32: xfused_n<32>(a, nouter, ncomp, stream_); break;
    case 64: xfused_n<64>(a, nouter, ncomp, stream_); break;
    case 128: xfused_n<128>(a, nouter, ncomp, stream_); break;
    case 256: xfused_n<256>(a, nouter, ncomp, stream_); break;
    case 512: xfused_n<512>(a, nouter, ncomp, stream_); break;
    case 1024: xfused_n<1024>(a, nouter, ncomp, stream_); break;
    default: throw std::runtime_error("fft: unsupported fused length");
  }
}

bool Fft3::can_block_y(int nranks) const {
  if (nranks < 1 || g_.ny % nranks) return false;
  const int nyl = g_.ny / nranks;
  return fast_[1] && is_pow2(nyl);
}

void Fft3::c2c_y_blocked(double* in, long in_cs, double* out, long out_cs, int ncomp, int dir, double scale, int nranks,
                         int interleave) {
  if (!can_block_y(nranks)) throw std::runtime_error("fft: blocked y pass not available for this length");
  const int nyl = g_.ny / nranks;
  int sh = 0;
  while ((1 << sh) < nyl) ++sh;
  const long plain_os = (long)g_.ny * g_.nzc, blocked_os = (long)nyl * g_.nzc;
  // interleave = 3: a peer's block holds its three components one after the other ([q][c][nx][ny/P][nzc]; the component
  // stride of the blocked side is then ONE block) -- one message per peer
  const long jump = (long)interleave * g_.nx * nyl * g_.nzc - (long)nyl * g_.nzc;
  StridedArgs a;
  a.data = reinterpret_cast<cplx*>(in);
  a.out = reinterpret_cast<cplx*>(out);
  a.out_cs = out_cs / 2;
  a.ls = g_.nzc;
  a.ncols = g_.nzc;
  a.tiles_per_outer = 0;
  a.scale = scale;
  a.tw = tw_[1];
  a.nt = stream_stores_ ? (1 | ((nt_loads_env() & 1) ? 2 : 0)) : 0;
  a.xcd_order = 0;
  if (dir < 0) {
    a.os = plain_os;
    a.os_out = blocked_os;
    a.split_out = sh;
    a.jump_out = jump;
  } else {
    a.os = blocked_os;
    a.os_out = plain_os;
    a.split_in = sh;
    a.jump_in = jump;
  }
  strided_pow2(g_.ny, a, g_.nx, dir, ncomp, in_cs / 2, stream_);
}

void Fft3::c2c_y(double* data, int ncomp, long comp_stride, int dir, double scale, const PlaneWindow* w) {
  strided(data, ncomp, comp_stride, 1, dir, scale, w);
}
void Fft3::c2c_x(double* data, int ncomp, long comp_stride, int dir, double scale) {
  strided(data, ncomp, comp_stride, 0, dir, scale);
}

void Fft3::r2c_z(double* data, int ncomp, long comp_stride, const PlaneWindow* w) {
  long nrows = (long)g_.nx * g_.ny;
  if (w && w->np >= 0) {
    if (!fast_[2] || w->x0 < 0 || w->x0 + w->np > g_.nx) throw std::runtime_error("fft: plane windows need a power-of-two z pass");
    data += (long)w->x0 * g_.ny * g_.nzp;
    nrows = (long)w->np * g_.ny;
    if (nrows == 0) return;
  }
  if (fast_[2]) {
    ZArgs a = {data, nrows, g_.nzp, tw_[2], wz_, stream_stores_ ? (1 | ((nt_loads_env() & 4) ? 2 : 0)) : 0};
    if (w && w->nt >= 0) a.nt = w->nt;
    // the real split right after the last pass, mirrored values by wave shuffle instead of a round trip of the spectrum
    // through LDS (R2CKernel<.., MIRROR>): 512^3 1.30 -> 1.16 ms, 256^3 0.158 -> 0.155 ms
    {
      switch (g_.nz / 2) {
#define FG_CASE(m) case m: launch_z<R2CKernel<m, ZLines<m>::value, true>>(a, ncomp, comp_stride, ZLines<m>::value, stream_); return;
        FG_CASE(64) FG_CASE(128) FG_CASE(256) FG_CASE(512)
#undef FG_CASE
        default: break;
      }
    }
    switch (g_.nz / 2) {
#define FG_CASE(m) case m: launch_z<R2CKernel<m, ZLines<m>::value>>(a, ncomp, comp_stride, ZLines<m>::value, stream_); break;
      FG_CASE(8) FG_CASE(16) FG_CASE(32) FG_CASE(64) FG_CASE(128) FG_CASE(256) FG_CASE(512) FG_CASE(1024)
#undef FG_CASE
      default: throw std::runtime_error("fft: unsupported fast z length");
    }
    return;
  }
  if (smooth_[2].n) {
    SmoothZArgs a = {data, nrows, g_.nzp, wgen_[2], stream_stores_ ? 3 : 0, smooth_[2], zodd_ ? 1 : 0};
    launch_smooth_z(a, true, ncomp, comp_stride, stream_);
    return;
  }
  if (odd_[2]) {
    // nz = 2 M, M = p m: m-point kernels on the sub-rows of the packed real rows ([m][p] complex, line stride p), then
    // combine + real split per row through the scratch component
    const int M = g_.nz / 2, p = odd_[2], m = M / p;
    if ((p == 3 && z_mixed_p<3>(m, data, nrows, g_.nzp, ncomp, comp_stride, true, tw_[2], wgen_[2], stream_)) ||
        (p == 5 && z_mixed_p<5>(m, data, nrows, g_.nzp, ncomp, comp_stride, true, tw_[2], wgen_[2], stream_)) ||
        (p == 7 && z_mixed_p<7>(m, data, nrows, g_.nzp, ncomp, comp_stride, true, tw_[2], wgen_[2], stream_)) ||
        (p == 9 && z_mixed_p<9>(m, data, nrows, g_.nzp, ncomp, comp_stride, true, tw_[2], wgen_[2], stream_)))
      return;
    StridedArgs a;
    a.data = reinterpret_cast<cplx*>(data);
    a.ls = p;
    a.os = g_.nzc;
    a.ncols = p;
    a.tiles_per_outer = 0;
    a.scale = 1.0;
    a.tw = tw_[2];
    a.nt = 0;
    a.xcd_order = 0;
    strided_pow2_narrow(m, a, (int)nrows, -1, ncomp, comp_stride / 2, stream_);
    if ((size_t)(M + 1) * sizeof(cplx) > 144 * 1024) throw std::runtime_error("fft: z length too large for the sub-line path");
    int rows = (int)(48 * 1024 / ((M + 1) * sizeof(cplx)));   // rows per workgroup: 48 KB of LDS
    if (rows < 1) rows = 1;
    const size_t lds = (size_t)rows * (M + 1) * sizeof(cplx);
    const unsigned nb = (unsigned)((nrows + rows - 1) / rows);
    for (int c = 0; c < ncomp; ++c) {
      double* src = data + c * comp_stride;
#define FG_FINISH(PP)                                                                                             \
  do {                                                                                                            \
    static PerDeviceOnce configured;                                                                               \
    if (auto once = configured.first_use()) {                                                                                            \
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mixed_r2c_finish<PP>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                  \
    }                                                                                                             \
    hipLaunchKernelGGL(k_mixed_r2c_finish<PP>, dim3(nb), dim3(256), lds, stream_, src, nrows, g_.nzp, M, p, rows, \
                       wgen_[2]);                                                                                 \
  } while (0)
      if (p == 3) FG_FINISH(3);
      else if (p == 5) FG_FINISH(5);
      else if (p == 7) FG_FINISH(7);
      else if (p == 9) FG_FINISH(9);
      else if (p == 15) FG_FINISH(15);
      else if (p == 25) FG_FINISH(25);
      else FG_FINISH(0);
#undef FG_FINISH
      FG_HIP_CHECK(hipGetLastError());
    }
    return;
  }
  const long total = nrows * g_.nzc;
  const int bs = 256;
  for (int c = 0; c < ncomp; ++c) {
    double* src = data + c * comp_stride;
    hipLaunchKernelGGL(k_r2c_generic, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, stream_, src, scratch_,
                       nrows, g_.nz, g_.nzp, wgen_[2]);
    FG_HIP_CHECK(hipGetLastError());
    FG_HIP_CHECK(hipMemcpyAsync(src, scratch_, g_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  }
}

void Fft3::c2r_z(double* data, int ncomp, long comp_stride, const PlaneWindow* w) {
  long nrows = (long)g_.nx * g_.ny;
  if (w && w->np >= 0) {
    if (!fast_[2] || w->x0 < 0 || w->x0 + w->np > g_.nx) throw std::runtime_error("fft: plane windows need a power-of-two z pass");
    data += (long)w->x0 * g_.ny * g_.nzp;
    nrows = (long)w->np * g_.ny;
    if (nrows == 0) return;
  }
  if (fast_[2]) {
    ZArgs a = {data, nrows, g_.nzp, tw_[2], wz_, stream_stores_ ? (1 | ((nt_loads_env() & 8) ? 2 : 0)) : 0};
    if (w && w->nt >= 0) a.nt = w->nt;
    // every coefficient read once: the mirrored one comes from the neighbouring lane (C2RKernel<.., MIRROR>); 256^3
    // 0.164 -> 0.151 ms, 512^3 1.38 -> 1.19 ms
    {
      switch (g_.nz / 2) {
#define FG_CASE(m) case m: launch_z<C2RKernel<m, ZLines<m>::value, true>>(a, ncomp, comp_stride, ZLines<m>::value, stream_); return;
        FG_CASE(64) FG_CASE(128) FG_CASE(256) FG_CASE(512)
#undef FG_CASE
        default: break;
      }
    }
    switch (g_.nz / 2) {
#define FG_CASE(m) case m: launch_z<C2RKernel<m, ZLines<m>::value>>(a, ncomp, comp_stride, ZLines<m>::value, stream_); break;
      FG_CASE(8) FG_CASE(16) FG_CASE(32) FG_CASE(64) FG_CASE(128) FG_CASE(256) FG_CASE(512) FG_CASE(1024)
#undef FG_CASE
      default: throw std::runtime_error("fft: unsupported fast z length");
    }
    return;
  }
  if (smooth_[2].n) {
    SmoothZArgs a = {data, nrows, g_.nzp, wgen_[2], stream_stores_ ? 3 : 0, smooth_[2], zodd_ ? 1 : 0};
    launch_smooth_z(a, false, ncomp, comp_stride, stream_);
    return;
  }
  if (odd_[2]) {
    const int M = g_.nz / 2, p = odd_[2], m = M / p;
    if ((p == 3 && z_mixed_p<3>(m, data, nrows, g_.nzp, ncomp, comp_stride, false, tw_[2], wgen_[2], stream_)) ||
        (p == 5 && z_mixed_p<5>(m, data, nrows, g_.nzp, ncomp, comp_stride, false, tw_[2], wgen_[2], stream_)) ||
        (p == 7 && z_mixed_p<7>(m, data, nrows, g_.nzp, ncomp, comp_stride, false, tw_[2], wgen_[2], stream_)) ||
        (p == 9 && z_mixed_p<9>(m, data, nrows, g_.nzp, ncomp, comp_stride, false, tw_[2], wgen_[2], stream_)))
      return;
    int rows = (int)(48 * 1024 / ((M + 1) * sizeof(cplx)));
    if (rows < 1) rows = 1;
    const size_t lds = (size_t)rows * (M + 1) * sizeof(cplx);
    const unsigned nb = (unsigned)((nrows + rows - 1) / rows);
    for (int c = 0; c < ncomp; ++c) {
      double* src = data + c * comp_stride;
#define FG_START(PP)                                                                                             \
  do {                                                                                                           \
    static PerDeviceOnce configured;                                                                              \
    if (auto once = configured.first_use()) {                                                                                           \
      FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mixed_c2r_start<PP>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
    }                                                                                                            \
    hipLaunchKernelGGL(k_mixed_c2r_start<PP>, dim3(nb), dim3(256), lds, stream_, src, nrows, g_.nzp, M, p, rows, \
                       wgen_[2]);                                                                                \
  } while (0)
      if (p == 3) FG_START(3);
      else if (p == 5) FG_START(5);
      else if (p == 7) FG_START(7);
      else if (p == 9) FG_START(9);
      else if (p == 15) FG_START(15);
      else if (p == 25) FG_START(25);
      else FG_START(0);
#undef FG_START
      FG_HIP_CHECK(hipGetLastError());
    }
    StridedArgs a;
    a.data = reinterpret_cast<cplx*>(data);
    a.ls = p;
    a.os = g_.nzc;
    a.ncols = p;
    a.tiles_per_outer = 0;
    a.scale = 1.0;
    a.tw = tw_[2];
    a.nt = 0;
    a.xcd_order = 0;
    strided_pow2_narrow(m, a, (int)nrows, +1, ncomp, comp_stride / 2, stream_);
    return;
  }
  const long total = nrows * g_.nz;
  const int bs = 256;
  for (int c = 0; c < ncomp; ++c) {
    double* src = data + c * comp_stride;
    hipLaunchKernelGGL(k_c2r_generic, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, stream_, src, scratch_,
                       nrows, g_.nz, g_.nzp, wgen_[2]);
    FG_HIP_CHECK(hipGetLastError());
    FG_HIP_CHECK(hipMemcpyAsync(src, scratch_, g_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  }
}

void Fft3::forward(double* data, int ncomp, long comp_stride, double scale) {
  r2c_z(data, ncomp, comp_stride);
  // fold the scale into the last pass that actually runs
  const bool has_x = g_.nx > 1, has_y = g_.ny > 1;
  c2c_y(data, ncomp, comp_stride, -1, (has_x || !has_y) ? 1.0 : scale);
  c2c_x(data, ncomp, comp_stride, -1, has_x ? scale : 1.0);
  if (!has_x && !has_y && scale != 1.0) this->scale(data, ncomp, comp_stride, scale);
}

// explicit scaling sweep (only for 1-D problems along z, where no strided pass can carry the factor)
void Fft3::scale(double* data, int ncomp, long comp_stride, double scale) {
  const int bs = 256;
  for (int c = 0; c < ncomp; ++c) {
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((g_.n + bs - 1) / bs)), dim3(bs), 0, stream_, data + c * comp_stride, g_.n,
                       scale);
    FG_HIP_CHECK(hipGetLastError());
  }
}

void Fft3::inverse(double* data, int ncomp, long comp_stride) {
  c2c_x(data, ncomp, comp_stride, +1, 1.0);
  c2c_y(data, ncomp, comp_stride, +1, 1.0);
  c2r_z(data, ncomp, comp_stride);
}

}  // namespace fg
