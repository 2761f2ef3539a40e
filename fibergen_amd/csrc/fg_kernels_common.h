// Device helpers shared by the kernel translation units (fg_kernels.hip: exact operation order,
// no FMA contraction; fg_kernels_fast.hip: precomputed moduli, FMA contraction).
#pragma once

#include <stdexcept>

#include "fg_kernels.h"

namespace fg {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ double2 ld2(const double* p, long i) { return *reinterpret_cast<const double2*>(p + i); }
__device__ __forceinline__ void st2(double* p, long i, double2 v) { *reinterpret_cast<double2*>(p + i) = v; }

// Decompose a pair index into (row, k): rows are (i,j) lines of nzc pairs.
struct PairPos {
  long row;   // i*ny + j
  int i, j, k;
  long off;   // element offset of (i,j,k)
};

__device__ __forceinline__ PairPos pair_pos(long pidx, const Grid& g) {
  PairPos p;
  p.row = pidx / g.nzc;
  p.k = 2 * (int)(pidx - p.row * g.nzc);
  p.i = (int)(p.row / g.ny);
  p.j = (int)(p.row - (long)p.i * g.ny);
  p.off = p.row * g.nzp + p.k;
  return p;
}

// L2-aware traversal for the stencil kernels.  The pair space is re-ordered as
// [y-chunk of ry rows][x][row in chunk][z pair] and every block sweeps one contiguous run of it;
// blocks that share an XCD (blockIdx % 8, guide section 1) get adjacent runs.  The +-1 neighbours in
// y (same chunk) and in x (next step of the sweep, ry rows x 9 arrays ~ 300 KB apart) are then served
// by that XCD's 4 MiB L2 instead of being fetched again: the row-major sweep read every neighbour
// from HBM (FETCH_SIZE 2.1x / 3.4x the algorithmic bytes for div / eps, profiles/r01_pmc_*).
struct BlockRun {
  long first, stride, count;  // pieces of kBlock pairs: first, first + stride, ...
};

// Block b works on piece remap(b) of kBlock pairs (sweep_blocks launches at least one block per piece):
// at any time the resident blocks work on one contiguous window of the re-ordered pair space.  (Blocks
// that owned a long contiguous run of their own made the XCD's L2 juggle hundreds of far-apart streams --
// measured: no reuse at all; blocks looping over strided pieces drifted out of step, 2-3x over-fetch.)
__device__ __forceinline__ BlockRun block_run(long npieces) {
  const unsigned nb = gridDim.x, b = blockIdx.x;
  BlockRun r;
  r.first = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
  r.stride = nb;
  r.count = r.first < npieces ? 1 : 0;
  return r;
}

// Division by a run-time constant without the 64-bit divide sequence: for 0 <= n < 2^31 and
// l = ceil(log2 d), m = ceil(2^(31+l) / d) < 2^32 gives n / d == (n * m) >> (31 + l) exactly
// (Granlund-Montgomery round-up; the error term n e / (d 2^(31+l)) stays below 2^-l <= 1/d).
struct FastDiv {
  unsigned d, mul, sh;
};
inline FastDiv make_fastdiv(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  const unsigned long long p = 1ull << (31 + l);
  return FastDiv{d, (unsigned)((p + d - 1) / d), 31 + l};
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) {
  return (unsigned)(((unsigned long long)n * f.mul) >> f.sh);
}

// parameters of the L2-aware sweep: rows per y-chunk (a power of two) and the dividers of its index map
struct Sweep {
  FastDiv by_nzc, by_nx;
  int ry, ry_shift;
};

__device__ __forceinline__ PairPos pair_pos_tiled(long q, const Grid& g, const Sweep& sw) {
  PairPos p;
  unsigned t = fast_div((unsigned)q, sw.by_nzc);  // launches guarantee npairs < 2^31
  p.k = 2 * (int)((unsigned)q - t * (unsigned)g.nzc);
  const int jr = (int)(t & (unsigned)(sw.ry - 1));
  t >>= sw.ry_shift;
  const unsigned jc = fast_div(t, sw.by_nx);
  p.i = (int)(t - jc * (unsigned)g.nx);
  p.j = (int)jc * sw.ry + jr;
  p.row = (long)p.i * g.ny + p.j;
  p.off = p.row * g.nzp + p.k;
  return p;
}

// Lane permutations as DPP moves (VALU only, no LDS traffic).  CTRL: 0x120+n = row_ror:n (rotate
// within each row of 16 lanes), 0x138 = wave_shr:1 (lane i <- i-1), 0x130 = wave_shl:1 (lane i <- i+1).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Deterministic block reduction of NV values per thread: fixed rotate tree inside each row of 16
// lanes, the four row totals and then the four waves combined in a fixed order; thread 0 holds the result.
template <int NV, class Op>
__device__ __forceinline__ void block_reduce(double* v, double* smem, Op op) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double a = v[q];
    a = op(a, dpp_move<0x128>(a));
    a = op(a, dpp_move<0x124>(a));
    a = op(a, dpp_move<0x122>(a));
    a = op(a, dpp_move<0x121>(a));
    v[q] = op(op(read_lane(a, 0), read_lane(a, 16)), op(read_lane(a, 32), read_lane(a, 48)));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int nw = kBlock / 64;
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) smem[wave * NV + q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      double a = smem[q];
#pragma unroll
      for (int w = 1; w < nw; ++w) a = op(a, smem[w * NV + q]);
      v[q] = a;
    }
  }
}

struct OpSum { __device__ double operator()(double a, double b) const { return a + b; } };
struct OpMin { __device__ double operator()(double a, double b) const { return a < b ? a : b; } };
struct OpMax { __device__ double operator()(double a, double b) const { return a > b ? a : b; } };

struct Row4 {
  double v[4];
};

__device__ __forceinline__ Row4 load_row(const double* a, long ro, int k, int kb, int kf2, bool second, bool m1, bool p2) {
  Row4 r;
  const double2 d = ld2(a, ro + k);
  r.v[1] = d.x;
  r.v[2] = second ? d.y : a[ro];  // odd nz, last pair: k+1 wraps to 0
  r.v[0] = m1 ? a[ro + kb] : 0.0;
  r.v[3] = p2 ? a[ro + kf2] : 0.0;
  return r;
}

// second stage: one block folds nblocks x NV partials in a fixed order
template <class Op>
__global__ __launch_bounds__(kBlock) void k_fold(const double* partial, int nblocks, int nv, double init, double* out) {
  __shared__ double smem[kBlock];
  Op op;
  for (int q = 0; q < nv; ++q) {
    double a = init;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) a = op(a, partial[(long)b * nv + q]);
    smem[threadIdx.x] = a;
    __syncthreads();
    for (int s = kBlock / 2; s >= 1; s >>= 1) {
      if ((int)threadIdx.x < s) smem[threadIdx.x] = op(smem[threadIdx.x], smem[threadIdx.x + s]);
      __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = smem[0];
    __syncthreads();
  }
}


// blocks of a stencil sweep: one 256-pair piece per block (rounded up to a multiple of 8 so that the
// XCD remap applies; surplus blocks find no piece).  Blocks that each looped over many pieces drifted
// out of step and lost the neighbour reuse in L2 (measured 2-3x over-fetch), hence one piece per block.
inline int sweep_blocks(long npairs) {
  long b = (npairs + kBlock - 1) / kBlock;
  if (b >= 8) b = ((b + 7) / 8) * 8;
  return (int)(b < 1 ? 1 : b);
}

// fold level for long partial lists: block i sums rows [i*rows_per, (i+1)*rows_per) in order
template <class Op>
__global__ __launch_bounds__(kBlock) void k_fold_level(const double* partial, long nrows, int rows_per, int nv, double init,
                                                       double* out) {
  __shared__ double smem[kBlock];
  Op op;
  const long r0 = (long)blockIdx.x * rows_per;
  const long r1 = r0 + rows_per < nrows ? r0 + rows_per : nrows;
  for (int q = 0; q < nv; ++q) {
    double a = init;
    for (long b = r0 + threadIdx.x; b < r1; b += blockDim.x) a = op(a, partial[b * nv + q]);
    smem[threadIdx.x] = a;
    __syncthreads();
    for (int s = kBlock / 2; s >= 1; s >>= 1) {
      if ((int)threadIdx.x < s) smem[threadIdx.x] = op(smem[threadIdx.x], smem[threadIdx.x + s]);
      __syncthreads();
    }
    if (threadIdx.x == 0) out[(long)blockIdx.x * nv + q] = smem[0];
    __syncthreads();
  }
}

// deterministic sum of nrows x nv partials into out[nv]; lists longer than 8192 rows go through one
// intermediate level stored behind the partials
inline void fold_sum(double* partial, long nrows, int nv, double* out, hipStream_t s) {
  if (nrows <= 8192) {
    hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, partial, (int)nrows, nv, 0.0, out);
    return;
  }
  const int rows_per = 512;
  const int nb = (int)((nrows + rows_per - 1) / rows_per);
  double* mid = partial + nrows * nv;
  hipLaunchKernelGGL(k_fold_level<OpSum>, dim3(nb), dim3(kBlock), 0, s, partial, nrows, rows_per, nv, 0.0, mid);
  hipLaunchKernelGGL(k_fold<OpSum>, dim3(1), dim3(kBlock), 0, s, mid, nb, nv, 0.0, out);
}

// rows per y-chunk of the L2-aware sweep
inline Sweep chunk_rows(const Grid& g) {
  if ((long)g.nx * g.ny * g.nzc >= (1L << 31)) throw std::runtime_error("grid too large for the 31-bit pair index of the sweeps");
  Sweep sw;
  sw.ry = g.ny % 8 == 0 ? 8 : (g.ny % 4 == 0 ? 4 : (g.ny % 2 == 0 ? 2 : 1));
  sw.ry_shift = sw.ry == 8 ? 3 : (sw.ry == 4 ? 2 : (sw.ry == 2 ? 1 : 0));
  sw.by_nzc = make_fastdiv((unsigned)g.nzc);
  sw.by_nx = make_fastdiv((unsigned)g.nx);
  return sw;
}

}  // namespace
}  // namespace fg
