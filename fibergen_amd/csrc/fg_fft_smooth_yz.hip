// Stockham tile kernels for lengths with factors 2, 3, 5, 7, 11, 13 (fg_fft_smooth.h): the strided (y / x) passes and the z passes.
#include "fg_fft_smooth_dev.h"

namespace fg {
namespace fft {

namespace {

// THREADS / RMAX classes: (256, 16), (256, 32), (1024, 16) -- registers follow the largest butterfly a kernel is built for
// CW: the tile width as a compile-time constant (8, 16, 32; 0 = the 4- / 2-column tiles of very long lines, run-time) -- as run-time
// branches of ONE kernel their load variants raised the allocation of the R <= 16 class from 127 to 147 VGPRs and cost the
// 8-column passes a resident workgroup (300^3 y passes 367 -> 470 us, old / new library alternating on one box; holding the
// kernel to 128 VGPRs by attribute made the scheduler serialise the loads instead: 438 us, and the z kernels spill)
template <int DIR, int THREADS, int RMAX, int CW>
__global__ __launch_bounds__(THREADS) void k_smooth_strided(SmoothArgs a, long comp_stride) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  a.data += (long)blockIdx.y * comp_stride;
  constexpr int B = THREADS == 256 ? 16 : 8;
  if constexpr (CW != 0) {
    smooth_strided_load<CW, B>(a, blockIdx.x, threadIdx.x, THREADS, img);
  } else {
    if (a.plan.lines == 4) smooth_strided_load<4, B>(a, blockIdx.x, threadIdx.x, THREADS, img);
    else smooth_strided_load<2, B>(a, blockIdx.x, threadIdx.x, THREADS, img);
  }
  __syncthreads();
  const SmoothMap L = {a.plan.lines, 1, a.plan.lines, false};
  smooth_dev_passes<DIR, RMAX>(img, a.plan, L, a.w, 1);
  if constexpr (CW != 0) {
    smooth_strided_store<CW>(a, blockIdx.x, threadIdx.x, THREADS, img);
  } else {
    if (a.plan.lines == 4) smooth_strided_store<4>(a, blockIdx.x, threadIdx.x, THREADS, img);
    else smooth_strided_store<2>(a, blockIdx.x, threadIdx.x, THREADS, img);
  }
}

template <bool FWD, int THREADS, int RMAX>
__global__ __launch_bounds__(THREADS) void k_smooth_z(SmoothZArgs a, long comp_stride) {
  extern __shared__ __align__(16) double lds[];
  cplx* img = reinterpret_cast<cplx*>(lds);
  a.data += (long)blockIdx.y * comp_stride;
  const long row0 = (long)blockIdx.x * a.plan.lines;
  const SmoothMap L = smooth_z_map(a.plan.n, a.plan.lines);
  constexpr int B = THREADS == 256 ? 16 : 8;
  if (a.odd) {   // odd nz: the row as nz complex points
    if (FWD) smooth_zodd_load_real<B>(a, row0, threadIdx.x, THREADS, img);
    else smooth_zodd_load_half<B>(a, row0, threadIdx.x, THREADS, img);
    __syncthreads();
    smooth_dev_passes<FWD ? -1 : +1, RMAX>(img, a.plan, L, a.w, 1);
    if (FWD) smooth_zodd_store_half(a, row0, threadIdx.x, THREADS, img);
    else smooth_zodd_store_real(a, row0, threadIdx.x, THREADS, img);
    return;
  }
  if (FWD) {
    smooth_z_load_packed<B>(a, row0, threadIdx.x, THREADS, img);
  } else {
    smooth_z_load_spectrum<B>(a, row0, threadIdx.x, THREADS, img);
    __syncthreads();
    smooth_z_merge<B>(a, threadIdx.x, THREADS, img);
  }
  __syncthreads();
  smooth_dev_passes<FWD ? -1 : +1, RMAX>(img, a.plan, L, a.w, 2);
  if (FWD) smooth_z_split_store<B>(a, row0, threadIdx.x, THREADS, img);
  else smooth_z_store_packed(a, row0, threadIdx.x, THREADS, img);
}

}  // namespace

void launch_smooth_strided(const SmoothArgs& a0, int nouter, int dir, int ncomp, long cs, hipStream_t s) {
  SmoothArgs a = a0;
  const int C = a.plan.lines;
  a.tiles_per_outer = (a.ncols + C - 1) / C;
  const size_t lds = (size_t)a.plan.n * C * sizeof(cplx);
  static PerDeviceOnce configured;
  if (auto once = configured.first_use()) {
    smooth_configure(&k_smooth_strided<-1, 256, 16, 0>);
    smooth_configure(&k_smooth_strided<+1, 256, 16, 0>);
    smooth_configure(&k_smooth_strided<-1, 256, 16, 8>);
    smooth_configure(&k_smooth_strided<+1, 256, 16, 8>);
    smooth_configure(&k_smooth_strided<-1, 256, 32, 8>);
    smooth_configure(&k_smooth_strided<+1, 256, 32, 8>);
    smooth_configure(&k_smooth_strided<-1, 1024, 16, 8>);
    smooth_configure(&k_smooth_strided<+1, 1024, 16, 8>);
    smooth_configure(&k_smooth_strided<-1, 256, 16, 16>);
    smooth_configure(&k_smooth_strided<+1, 256, 16, 16>);
    smooth_configure(&k_smooth_strided<-1, 256, 16, 32>);
    smooth_configure(&k_smooth_strided<+1, 256, 16, 32>);
    smooth_configure(&k_smooth_strided<-1, 256, 32, 0>);
    smooth_configure(&k_smooth_strided<+1, 256, 32, 0>);
    smooth_configure(&k_smooth_strided<-1, 1024, 16, 0>);
    smooth_configure(&k_smooth_strided<+1, 1024, 16, 0>);
  }
  const dim3 grid((unsigned)((long)a.tiles_per_outer * nouter), ncomp);
  if (launch_smooth_strided_plan(a, grid, lds, dir, cs, s)) return;
  const int cls = smooth_class(a.plan);
  if (C > 8 && cls != 0) throw std::runtime_error("fft: wide tiles are planned with radices <= 16 and 256 threads");
#define FG_GO(D, T, R, W) hipLaunchKernelGGL((k_smooth_strided<D, T, R, W>), grid, dim3(T), lds, s, a, cs)
  if (dir < 0) {
    if (C == 16) FG_GO(-1, 256, 16, 16);
    else if (C == 32) FG_GO(-1, 256, 16, 32);
    else if (C == 8 && cls == 0) FG_GO(-1, 256, 16, 8);
    else if (C == 8 && cls == 1) FG_GO(-1, 256, 32, 8);
    else if (C == 8) FG_GO(-1, 1024, 16, 8);
    else if (cls == 0) FG_GO(-1, 256, 16, 0);
    else if (cls == 1) FG_GO(-1, 256, 32, 0);
    else FG_GO(-1, 1024, 16, 0);
  } else {
    if (C == 16) FG_GO(+1, 256, 16, 16);
    else if (C == 32) FG_GO(+1, 256, 16, 32);
    else if (C == 8 && cls == 0) FG_GO(+1, 256, 16, 8);
    else if (C == 8 && cls == 1) FG_GO(+1, 256, 32, 8);
    else if (C == 8) FG_GO(+1, 1024, 16, 8);
    else if (cls == 0) FG_GO(+1, 256, 16, 0);
    else if (cls == 1) FG_GO(+1, 256, 32, 0);
    else FG_GO(+1, 1024, 16, 0);
  }
#undef FG_GO
  FG_HIP_CHECK(hipGetLastError());
}

void launch_smooth_z(const SmoothZArgs& a, bool fwd, int ncomp, long comp_stride, hipStream_t s) {
  const int lines = a.plan.lines;
  const size_t lds = (size_t)lines * smooth_z_pitch(a.plan.n) * sizeof(cplx);
  static PerDeviceOnce configured;
  if (auto once = configured.first_use()) {
    smooth_configure(&k_smooth_z<true, 256, 16>);
    smooth_configure(&k_smooth_z<false, 256, 16>);
    smooth_configure(&k_smooth_z<true, 256, 32>);
    smooth_configure(&k_smooth_z<false, 256, 32>);
    smooth_configure(&k_smooth_z<true, 1024, 16>);
    smooth_configure(&k_smooth_z<false, 1024, 16>);
  }
  const dim3 grid((unsigned)((a.nrows + lines - 1) / lines), ncomp);
  if (launch_smooth_z_plan(a, grid, lds, fwd, comp_stride, s)) return;
  switch (smooth_class(a.plan) * 2 + (fwd ? 0 : 1)) {
    case 0: hipLaunchKernelGGL((k_smooth_z<true, 256, 16>), grid, dim3(256), lds, s, a, comp_stride); break;
    case 1: hipLaunchKernelGGL((k_smooth_z<false, 256, 16>), grid, dim3(256), lds, s, a, comp_stride); break;
    case 2: hipLaunchKernelGGL((k_smooth_z<true, 256, 32>), grid, dim3(256), lds, s, a, comp_stride); break;
    case 3: hipLaunchKernelGGL((k_smooth_z<false, 256, 32>), grid, dim3(256), lds, s, a, comp_stride); break;
    case 4: hipLaunchKernelGGL((k_smooth_z<true, 1024, 16>), grid, dim3(1024), lds, s, a, comp_stride); break;
    default: hipLaunchKernelGGL((k_smooth_z<false, 1024, 16>), grid, dim3(1024), lds, s, a, comp_stride); break;
  }
  FG_HIP_CHECK(hipGetLastError());
}

}  // namespace fft
}  // namespace fg
