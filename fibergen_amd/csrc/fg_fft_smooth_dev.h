// Device side shared by the translation units of the Stockham tile kernels (fg_fft_smooth_yz.hip: strided and z passes;
// fg_fft_smooth_x.hip: fused x pass): the pass loop over the plan's radices, the LDS opt-in, the kernel class of a plan.
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "fg_fft_smooth.h"
#include "fg_hip_util.h"

namespace fg {
namespace fft {

// one pass of the tile in LDS: read + butterfly, barrier, write, barrier
template <int R, int DIR, int CAP, bool FDIV>
__device__ __forceinline__ void smooth_dev_pass(cplx* img, int N, int Ns, const SmoothMap& L, const cplx* w, int wscale) {
  constexpr int Q = CAP ? smooth_rounds(R, CAP) : 1;   // butterflies a thread may own (small radices: several)
  cplx v[Q][R];
  bool active[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) active[q] = smooth_pass_read<R, DIR, FDIV>(img, N, Ns, L, w, wscale, threadIdx.x + q * blockDim.x, v[q]);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (active[q]) smooth_pass_write<R, FDIV>(img, N, Ns, L, threadIdx.x + q * blockDim.x, v[q]);
  __syncthreads();
}

// RMAX: the largest radix the kernel is built for (32 with 256 threads, 16 with 1024: the butterfly lives in registers)
template <int DIR, int RMAX, int CAP = 20, bool FDIV = true>
__device__ __forceinline__ void smooth_dev_passes(cplx* img, const SmoothPlan& plan, const SmoothMap& L, const cplx* w, int wscale) {
  int Ns = 1;
  for (int f = 0; f < plan.npass; ++f) {
    // (the run-time index sends the by-value plan to scratch memory, 104 bytes -- and the radix into a vector register: with
    // static indices the switch below becomes uniform and the allocation of the (256, R <= 16) kernels goes from 149 to 247 VGPRs)
    const int R = plan.fac[f];
    switch (R) {
#define FG_R(r) case r: if constexpr (r <= RMAX) smooth_dev_pass<r, DIR, CAP, FDIV>(img, plan.n, Ns, L, w, wscale); break;
      FG_R(2) FG_R(3) FG_R(4) FG_R(5) FG_R(6) FG_R(7) FG_R(8) FG_R(9) FG_R(10) FG_R(11) FG_R(12) FG_R(13) FG_R(14) FG_R(15)
      FG_R(16) FG_R(18) FG_R(20) FG_R(21) FG_R(22) FG_R(24) FG_R(25) FG_R(26) FG_R(27) FG_R(28) FG_R(30) FG_R(32)
#undef FG_R
      default: break;
    }
    Ns *= R;
  }
}

template <class K>
void smooth_configure(K kernel) {
  FG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmoothLdsMax));
}

// the kernel class of a plan: 0 = (256 threads, radices <= 16), 1 = (256, <= 32), 2 = (1024, <= 16)
inline int smooth_class(const SmoothPlan& p) { return p.threads == 1024 ? 2 : (p.rmax() <= 16 && p.cap <= 20 ? 0 : 1); }


// launchers (fg_fft.hip calls them with the plans Fft3 made)
void launch_smooth_strided(const SmoothArgs& a0, int nouter, int dir, int ncomp, long cs, hipStream_t s);
void launch_smooth_z(const SmoothZArgs& a, bool fwd, int ncomp, long comp_stride, hipStream_t s);
void launch_smooth_xfused(const SmoothXArgs& a0, hipStream_t s);
// the kernels built for one plan each (fg_fft_smooth_plans.h); false: none for this plan, or switched off
bool launch_smooth_strided_plan(const SmoothArgs& a, const dim3& grid, size_t lds, int dir, long cs, hipStream_t s);
bool launch_smooth_z_plan(const SmoothZArgs& a, const dim3& grid, size_t lds, bool fwd, long cs, hipStream_t s);
bool launch_smooth_x_plan(const SmoothXArgs& a, const dim3& grid, size_t lds, hipStream_t s);

}  // namespace fft
}  // namespace fg
