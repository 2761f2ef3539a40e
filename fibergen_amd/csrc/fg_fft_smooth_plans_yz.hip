// Tile kernels built for one plan each: strided (y / x) passes and z passes (see fg_fft_smooth_plans.h).
#include <atomic>

#include "fg_fft_smooth_dev.h"
#include "fg_fft_smooth_plans.h"

namespace fg {
namespace fft {

namespace {

std::atomic<int> g_plan_kernels{1};

template <int DIR, int N, int CW, int R0, int R1, int R2>
__global__ __launch_bounds__(256) void k_smooth_strided_plan(SmoothArgs a, long comp_stride) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  a.data += (long)blockIdx.y * comp_stride;
  a.plan.n = N;
  a.plan.lines = CW;
  smooth_strided_load<CW, 16>(a, blockIdx.x, threadIdx.x, 256, img);
  __syncthreads();
  const SmoothMap L = {CW, 1, CW, false};
  smooth_dev_pass<R0, DIR, 20, false>(img, N, 1, L, a.w, 1);
  smooth_dev_pass<R1, DIR, 20, false>(img, N, R0, L, a.w, 1);
  if constexpr (R2 > 1) smooth_dev_pass<R2, DIR, 20, false>(img, N, R0 * R1, L, a.w, 1);
  smooth_strided_store<CW>(a, blockIdx.x, threadIdx.x, 256, img);
}

template <bool FWD, int M, int LINES, int R0, int R1, int R2>
__global__ __launch_bounds__(256) void k_smooth_z_plan(SmoothZArgs a, long comp_stride) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  a.data += (long)blockIdx.y * comp_stride;
  a.plan.n = M;
  a.plan.lines = LINES;
  const long row0 = (long)blockIdx.x * LINES;
  const SmoothMap L = smooth_z_map(M, LINES);
  constexpr int DIR = FWD ? -1 : +1;
  if (FWD) {
    smooth_z_load_packed<16>(a, row0, threadIdx.x, 256, img);
  } else {
    smooth_z_load_spectrum<16>(a, row0, threadIdx.x, 256, img);
    __syncthreads();
    smooth_z_merge<16>(a, threadIdx.x, 256, img);
  }
  __syncthreads();
  smooth_dev_pass<R0, DIR, 20, false>(img, M, 1, L, a.w, 2);
  smooth_dev_pass<R1, DIR, 20, false>(img, M, R0, L, a.w, 2);
  if constexpr (R2 > 1) smooth_dev_pass<R2, DIR, 20, false>(img, M, R0 * R1, L, a.w, 2);
  if (FWD) smooth_z_split_store<16>(a, row0, threadIdx.x, 256, img);
  else smooth_z_store_packed(a, row0, threadIdx.x, 256, img);
}

}  // namespace

void smooth_plan_kernels(bool on) { g_plan_kernels.store(on ? 1 : 0); }
bool smooth_plan_kernels_on() { return g_plan_kernels.load() != 0; }

// a: as launch_smooth_strided has prepared it (tiles_per_outer set); false: no kernel for this plan
bool launch_smooth_strided_plan(const SmoothArgs& a, const dim3& grid, size_t lds, int dir, long cs, hipStream_t s) {
  if (!smooth_plan_kernels_on()) return false;
#define FG_X(N, C, R0, R1, R2)                                                                                   \
  if (smooth_plan_is(a.plan, N, C, 256, 20, R0, R1, R2)) {                                                       \
    if (dir < 0) hipLaunchKernelGGL((k_smooth_strided_plan<-1, N, C, R0, R1, R2>), grid, dim3(256), lds, s, a, cs); \
    else hipLaunchKernelGGL((k_smooth_strided_plan<+1, N, C, R0, R1, R2>), grid, dim3(256), lds, s, a, cs);        \
    FG_HIP_CHECK(hipGetLastError());                                                                             \
    return true;                                                                                                 \
  }
  FG_SMOOTH_STRIDED_PLANS(FG_X)
#undef FG_X
  return false;
}

bool launch_smooth_z_plan(const SmoothZArgs& a, const dim3& grid, size_t lds, bool fwd, long cs, hipStream_t s) {
  if (!smooth_plan_kernels_on() || a.odd) return false;
#define FG_X(M, LINES, R0, R1, R2)                                                                               \
  if (smooth_plan_is(a.plan, M, LINES, 256, 20, R0, R1, R2)) {                                                   \
    if (fwd) hipLaunchKernelGGL((k_smooth_z_plan<true, M, LINES, R0, R1, R2>), grid, dim3(256), lds, s, a, cs);     \
    else hipLaunchKernelGGL((k_smooth_z_plan<false, M, LINES, R0, R1, R2>), grid, dim3(256), lds, s, a, cs);        \
    FG_HIP_CHECK(hipGetLastError());                                                                             \
    return true;                                                                                                 \
  }
  FG_SMOOTH_Z_PLANS(FG_X)
#undef FG_X
  return false;
}

}  // namespace fft
}  // namespace fg
