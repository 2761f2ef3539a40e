// Tile kernels built for one plan each: the fused x pass on the joint image of three components (see fg_fft_smooth_plans.h).
#include "fg_fft_smooth_dev.h"
#include "fg_fft_smooth_plans.h"

namespace fg {
namespace fft {

namespace {

template <int THREADS, int CAP, int N, int R0, int R1, int R2, int C>
__global__ __launch_bounds__(THREADS, 2) void k_smooth_xjoint_plan(SmoothXArgs a) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  constexpr int W = 3 * C;
  a.base.plan.n = N;
  smooth_joint_load<C, 3, 16>(a, blockIdx.x, threadIdx.x, THREADS, img);
  __syncthreads();
  const SmoothMap L = {W, 1, W, false};
  smooth_dev_pass<R0, -1, CAP, false>(img, N, 1, L, a.base.w, 1);
  smooth_dev_pass<R1, -1, CAP, false>(img, N, R0, L, a.base.w, 1);
  if constexpr (R2 > 1) smooth_dev_pass<R2, -1, CAP, false>(img, N, R0 * R1, L, a.base.w, 1);
  smooth_joint_green<C, 3>(a, blockIdx.x, threadIdx.x, THREADS, img);
  __syncthreads();
  smooth_dev_pass<R0, +1, CAP, false>(img, N, 1, L, a.base.w, 1);
  smooth_dev_pass<R1, +1, CAP, false>(img, N, R0, L, a.base.w, 1);
  if constexpr (R2 > 1) smooth_dev_pass<R2, +1, CAP, false>(img, N, R0 * R1, L, a.base.w, 1);
  smooth_joint_store<C, 3>(a, blockIdx.x, threadIdx.x, THREADS, img);
}

}  // namespace

// a: as launch_smooth_xfused has prepared it (tiles_per_outer set); false: no kernel for this plan
bool launch_smooth_x_plan(const SmoothXArgs& a, const dim3& grid, size_t lds, hipStream_t s) {
  if (!smooth_plan_kernels_on() || a.ncomp != 3 || a.base.plan.joint != 3) return false;
#define FG_X(N, C, T, CAP, R0, R1, R2)                                                                    \
  if (smooth_plan_is(a.base.plan, N, 3 * C, T, CAP, R0, R1, R2)) {                                        \
    static PerDeviceOnce configured;                                                                      \
    if (auto once = configured.first_use()) smooth_configure(&k_smooth_xjoint_plan<T, CAP, N, R0, R1, R2, C>); \
    hipLaunchKernelGGL((k_smooth_xjoint_plan<T, CAP, N, R0, R1, R2, C>), grid, dim3(T), lds, s, a);        \
    FG_HIP_CHECK(hipGetLastError());                                                                      \
    return true;                                                                                          \
  }
  FG_SMOOTH_X_PLANS(FG_X)
#undef FG_X
  return false;
}

}  // namespace fft
}  // namespace fg
