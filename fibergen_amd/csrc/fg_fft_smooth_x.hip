// Stockham tile kernels for lengths with factors 2, 3, 5, 7, 11, 13 (fg_fft_smooth.h): the fused x pass (x transform, 1/N,
// Green operator, inverse x transform of a tile in one kernel).
#include "fg_fft_smooth_dev.h"

namespace fg {
namespace fft {

namespace {

// x transform + Green operator + inverse x transform of the tile's NC components (NC images in LDS)
template <int THREADS, int RMAX, int NC>
__global__ __launch_bounds__(THREADS) void k_smooth_xfused(SmoothXArgs a) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  constexpr int B = THREADS == 256 ? 16 : 8;
  const int C = a.base.plan.lines;
  const long comp = (long)a.base.plan.n * C;
  const SmoothMap L = {C, 1, C, false};
#pragma nounroll   // (one component's batch of loads in registers at a time)
  for (int c = 0; c < NC; ++c) {
    SmoothArgs ac = a.base;
    ac.data += (long)c * a.comp_stride;
    if (C == 8) smooth_strided_load<8, B>(ac, blockIdx.x, threadIdx.x, THREADS, img + c * comp);
    else if (C == 16) smooth_strided_load<16, B>(ac, blockIdx.x, threadIdx.x, THREADS, img + c * comp);
    else smooth_strided_load<4, B>(ac, blockIdx.x, threadIdx.x, THREADS, img + c * comp);
  }
  __syncthreads();
#pragma nounroll
  for (int c = 0; c < NC; ++c) smooth_dev_passes<-1, RMAX, 0, false>(img + c * comp, a.base.plan, L, a.base.w, 1);
  if (C == 8) smooth_x_green<8, NC>(a, blockIdx.x, threadIdx.x, THREADS, img);
  else if (C == 16) smooth_x_green<16, NC>(a, blockIdx.x, threadIdx.x, THREADS, img);
  else smooth_x_green<4, NC>(a, blockIdx.x, threadIdx.x, THREADS, img);
  __syncthreads();
#pragma nounroll
  for (int c = 0; c < NC; ++c) smooth_dev_passes<+1, RMAX, 0, false>(img + c * comp, a.base.plan, L, a.base.w, 1);
#pragma nounroll
  for (int c = 0; c < NC; ++c) {
    SmoothArgs ac = a.base;
    ac.data += (long)c * a.comp_stride;
    ac.scale = 1.0;   // (the 1/N went in with the Green operator)
    if (C == 8) smooth_strided_store<8>(ac, blockIdx.x, threadIdx.x, THREADS, img + c * comp);
    else if (C == 16) smooth_strided_store<16>(ac, blockIdx.x, threadIdx.x, THREADS, img + c * comp);
    else smooth_strided_store<4>(ac, blockIdx.x, threadIdx.x, THREADS, img + c * comp);
  }
}

// The same on the joint image [kx][3][C] of the three components (SmoothPlan::joint): every pass runs once, over three times the
// butterflies -- the one-image-per-component form has 80 ... 160 of its 256 threads at work in a pass of a 200-point tile and
// twelve pass phases between its barriers; this one four.  Up to 20 values per thread (32 in the R <= 32 class): several
// butterflies of a small radix.  The register allocation follows the largest butterfly a kernel is BUILT for, not the plan's:
// (R <= 16) 228 VGPRs, (R <= 20) 272 -- one wave per SIMD -- held to 256 (30 spilled) for two: 200^3 fused x pass 326 -> 152 us;
// (R <= 32) 512 + scratch.  512 threads where the image leaves room for one workgroup per CU only (300, 360, 400 points).
template <int THREADS, int RMAX, int C, int NC>
__global__ __launch_bounds__(THREADS, RMAX <= 20 ? 2 : 1) void k_smooth_xjoint(SmoothXArgs a) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  constexpr int CAP = RMAX > 20 ? 32 : 20, W = NC * C;
  smooth_joint_load<C, NC, 16>(a, blockIdx.x, threadIdx.x, THREADS, img);
  __syncthreads();
  const SmoothMap L = {W, 1, W, false};
  smooth_dev_passes<-1, RMAX, CAP, false>(img, a.base.plan, L, a.base.w, 1);
  smooth_joint_green<C, NC>(a, blockIdx.x, threadIdx.x, THREADS, img);
  __syncthreads();
  smooth_dev_passes<+1, RMAX, CAP, false>(img, a.base.plan, L, a.base.w, 1);
  smooth_joint_store<C, NC>(a, blockIdx.x, threadIdx.x, THREADS, img);
}

}  // namespace

void launch_smooth_xfused(const SmoothXArgs& a0, hipStream_t s) {
  SmoothXArgs a = a0;
  const bool joint = a.base.plan.joint >= 1;
  const int C = joint ? a.base.plan.lines / a.base.plan.joint : a.base.plan.lines;
  a.base.tiles_per_outer = (a.base.ncols + C - 1) / C;
  const size_t lds = (size_t)a.ncomp * a.base.plan.n * C * sizeof(cplx);
  static PerDeviceOnce configured;
  if (auto once = configured.first_use()) {
    smooth_configure(&k_smooth_xfused<256, 16, 3>);
    smooth_configure(&k_smooth_xfused<256, 32, 3>);
    smooth_configure(&k_smooth_xfused<1024, 16, 3>);
    smooth_configure(&k_smooth_xfused<256, 16, 1>);
    smooth_configure(&k_smooth_xfused<256, 32, 1>);
    smooth_configure(&k_smooth_xfused<1024, 16, 1>);
    smooth_configure(&k_smooth_xjoint<512, 20, 4, 3>);
#define FG_CFG(NC) \
    smooth_configure(&k_smooth_xjoint<256, 16, 8, NC>); smooth_configure(&k_smooth_xjoint<256, 20, 8, NC>); \
    smooth_configure(&k_smooth_xjoint<512, 20, 8, NC>); smooth_configure(&k_smooth_xjoint<256, 32, 8, NC>); \
    smooth_configure(&k_smooth_xjoint<256, 16, 16, NC>); smooth_configure(&k_smooth_xjoint<256, 20, 16, NC>); \
    smooth_configure(&k_smooth_xjoint<512, 20, 16, NC>); smooth_configure(&k_smooth_xjoint<256, 32, 16, NC>)
    FG_CFG(3);
    FG_CFG(1);
#undef FG_CFG
  }
  const dim3 grid((unsigned)a.base.tiles_per_outer);
  if (joint && launch_smooth_x_plan(a, grid, lds, s)) return;
  if (joint) {
    const int rm = a.base.plan.rmax(), cap = a.base.plan.cap, T = a.base.plan.threads;
    if (a.ncomp != a.base.plan.joint || (C != 4 && C != 8 && C != 16) || (T != 256 && T != 512) || (T == 512 && (rm > 20 || cap > 20)) ||
        (C == 4 && (T != 512 || a.ncomp != 3)))
      throw std::runtime_error("fft: joint fused x pass: unsupported plan");
    const int k = T == 512 ? 2 : (rm <= 16 && cap <= 20 ? 0 : (rm <= 20 && cap <= 20 ? 1 : 3));
#define FG_GO1(TT, R, CC) \
  if (a.ncomp == 3) hipLaunchKernelGGL((k_smooth_xjoint<TT, R, CC, 3>), grid, dim3(TT), lds, s, a); \
  else hipLaunchKernelGGL((k_smooth_xjoint<TT, R, CC, 1>), grid, dim3(TT), lds, s, a)
#define FG_GO(TT, R) \
  if (C == 8) { FG_GO1(TT, R, 8); } else { FG_GO1(TT, R, 16); }
    if (C == 4) hipLaunchKernelGGL((k_smooth_xjoint<512, 20, 4, 3>), grid, dim3(512), lds, s, a);
    else if (k == 0) { FG_GO(256, 16) }
    else if (k == 1) { FG_GO(256, 20) }
    else if (k == 2) { FG_GO(512, 20) }
    else { FG_GO(256, 32) }
#undef FG_GO
#undef FG_GO1
    FG_HIP_CHECK(hipGetLastError());
    return;
  }
  const int cls = smooth_class(a.base.plan);
  switch (cls * 2 + (a.ncomp == 3 ? 0 : 1)) {
    case 0: hipLaunchKernelGGL((k_smooth_xfused<256, 16, 3>), grid, dim3(256), lds, s, a); break;
    case 1: hipLaunchKernelGGL((k_smooth_xfused<256, 16, 1>), grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL((k_smooth_xfused<256, 32, 3>), grid, dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL((k_smooth_xfused<256, 32, 1>), grid, dim3(256), lds, s, a); break;
    case 4: hipLaunchKernelGGL((k_smooth_xfused<1024, 16, 3>), grid, dim3(1024), lds, s, a); break;
    default: hipLaunchKernelGGL((k_smooth_xfused<1024, 16, 1>), grid, dim3(1024), lds, s, a); break;
  }
  FG_HIP_CHECK(hipGetLastError());
}


}  // namespace fft
}  // namespace fg
