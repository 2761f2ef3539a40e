// Device helpers shared by the kernel translation units (fg_kernels.hip: exact operation order,
// no FMA contraction; fg_kernels_fast.hip: precomputed moduli, FMA contraction).
#pragma once

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

// Piece r of block b is r*gridDim + remap(b): at any time the resident blocks work on one contiguous
// window of the re-ordered pair space (a block that owned a long contiguous run of its own would make
// the XCD's L2 juggle hundreds of far-apart streams -- measured: no reuse at all).
__device__ __forceinline__ BlockRun block_run(long npieces) {
  const long nb = gridDim.x, b = blockIdx.x;
  BlockRun r;
  r.first = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
  r.stride = nb;
  r.count = r.first < npieces ? (npieces - r.first + nb - 1) / nb : 0;
  return r;
}

__device__ __forceinline__ PairPos pair_pos_tiled(long q, const Grid& g, int ry) {
  PairPos p;
  long t = q / g.nzc;
  p.k = 2 * (int)(q - t * g.nzc);
  const int jr = (int)(t % ry);
  t /= ry;
  p.i = (int)(t % g.nx);
  const int jc = (int)(t / g.nx);
  p.j = jc * ry + jr;
  p.row = (long)p.i * g.ny + p.j;
  p.off = p.row * g.nzp + p.k;
  return p;
}

// Deterministic block reduction of NV values per thread: wave shuffle tree, then
// LDS across the 4 waves, lane 0 of wave 0 holds the result.
template <int NV, class Op>
__device__ __forceinline__ void block_reduce(double* v, double* smem, Op op) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v[q] = op(v[q], __shfl_down(v[q], s, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) smem[wave * NV + q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      double a = smem[q];
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
inline int chunk_rows(const Grid& g) { return g.ny % 8 == 0 ? 8 : (g.ny % 4 == 0 ? 4 : (g.ny % 2 == 0 ? 2 : 1)); }

}  // namespace
}  // namespace fg
