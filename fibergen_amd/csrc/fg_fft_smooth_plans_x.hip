// Tile kernels built for one plan each: the fused x pass on the joint image of three components / the one-component image of
// the scalar modes (see fg_fft_smooth_plans.h).
#include "fg_fft_smooth_dev.h"
#include "fg_fft_smooth_plans.h"

namespace fg {
namespace fft {

namespace {

template <int THREADS, int CAP, int N, int R0, int R1, int R2, int C, int NC>
__global__ __launch_bounds__(THREADS, 2) void k_smooth_xjoint_plan(SmoothXArgs a) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  constexpr int W = NC * C;
  // all loads of the tile in ONE batch where a thread has up to 24 of them (18.75 at 200 and 400 points: a second turn of the
  // batch loop with three loads exposed the memory latency once more)
  constexpr int PER = (N * W + THREADS - 1) / THREADS, B = PER > 16 && PER <= 24 ? PER : 16;
  a.base.plan.n = N;
  smooth_joint_load<C, NC, B>(a, blockIdx.x, threadIdx.x, THREADS, img);
  __syncthreads();
  const SmoothMap L = {W, 1, W, false};
  smooth_dev_pass<R0, -1, CAP, false>(img, N, 1, L, a.base.w, 1);
  smooth_dev_pass<R1, -1, CAP, false>(img, N, R0, L, a.base.w, 1);
  if constexpr (R2 > 1) smooth_dev_pass<R2, -1, CAP, false>(img, N, R0 * R1, L, a.base.w, 1);
  smooth_joint_green<C, NC>(a, blockIdx.x, threadIdx.x, THREADS, img);
  __syncthreads();
  smooth_dev_pass<R0, +1, CAP, false>(img, N, 1, L, a.base.w, 1);
  smooth_dev_pass<R1, +1, CAP, false>(img, N, R0, L, a.base.w, 1);
  if constexpr (R2 > 1) smooth_dev_pass<R2, +1, CAP, false>(img, N, R0 * R1, L, a.base.w, 1);
  smooth_joint_store<C, NC>(a, blockIdx.x, threadIdx.x, THREADS, img);
}

}  // namespace

// a: as launch_smooth_xfused has prepared it (tiles_per_outer set); false: no kernel for this plan
bool launch_smooth_x_plan(const SmoothXArgs& a, const dim3& grid, size_t lds, hipStream_t s) {
  if (!smooth_plan_kernels_on() || a.ncomp != a.base.plan.joint || (a.ncomp != 1 && a.ncomp != 3)) return false;
#define FG_GO(NC, N, C, T, CAP, R0, R1, R2)                                                                    \
  if (a.ncomp == NC && smooth_plan_is(a.base.plan, N, NC * C, T, CAP, R0, R1, R2)) {                           \
    static PerDeviceOnce configured;                                                                           \
    if (auto once = configured.first_use()) smooth_configure(&k_smooth_xjoint_plan<T, CAP, N, R0, R1, R2, C, NC>); \
    hipLaunchKernelGGL((k_smooth_xjoint_plan<T, CAP, N, R0, R1, R2, C, NC>), grid, dim3(T), lds, s, a);         \
    FG_HIP_CHECK(hipGetLastError());                                                                           \
    return true;                                                                                               \
  }
#define FG_X(N, C, T, CAP, R0, R1, R2) FG_GO(3, N, C, T, CAP, R0, R1, R2)
  FG_SMOOTH_X_PLANS(FG_X)
#undef FG_X
#define FG_X(N, C, T, CAP, R0, R1, R2) FG_GO(1, N, C, T, CAP, R0, R1, R2)
  FG_SMOOTH_X1_PLANS(FG_X)
#undef FG_X
#undef FG_GO
  return false;
}

}  // namespace fft
}  // namespace fg
