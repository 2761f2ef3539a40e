// Shared host/device helpers for the fibergen_amd kernels.
//
// Everything in this header is plain arithmetic that must behave identically
// on the device (hipcc, gfx950) and in the host-side emulation used by the
// CPU test-suite (g++, -DFG_HOST_EMULATION).  All floating point is IEEE
// float64 and the library is compiled with -ffp-contract=off so that the
// element-wise stages reproduce the reference's operation order bit for bit.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__) && !defined(FG_HOST_EMULATION)
#include <hip/hip_runtime.h>
#define FG_HD __host__ __device__ __forceinline__
#define FG_D __device__ __forceinline__
#else
#define FG_HD inline
#define FG_D inline
#endif

namespace fg {

struct alignas(16) cplx {
  double re, im;
};

FG_HD cplx cmake(double re, double im) { cplx c; c.re = re; c.im = im; return c; }
FG_HD cplx cadd(cplx a, cplx b) { return cmake(a.re + b.re, a.im + b.im); }
FG_HD cplx csub(cplx a, cplx b) { return cmake(a.re - b.re, a.im - b.im); }
// (a+ib)(c+id) = (ac - bd) + i(ad + bc): the finite-operand result of
// std::complex<double>::operator* (libstdc++ __muldc3) used by the reference.
FG_HD cplx cmul(cplx a, cplx b) { return cmake(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
FG_HD cplx cscale(double s, cplx a) { return cmake(s * a.re, s * a.im); }
FG_HD cplx cconj(cplx a) { return cmake(a.re, -a.im); }
// multiply by -i (forward) : (re,im) -> (im,-re) ; by +i : (re,im) -> (-im, re)
FG_HD cplx cmul_mi(cplx a) { return cmake(a.im, -a.re); }
FG_HD cplx cmul_pi(cplx a) { return cmake(-a.im, a.re); }

// Store of one complex value by an FFT pass.  The passes write every line exactly once and never read it back within
// the kernel; when the fields are larger than the Infinity Cache the store bypasses the cache allocation (nt != 0:
// 512^3 y pass -7 %, whole iteration 100 -> 105 it/s; 256^3 812 -> 863 it/s).  Small grids live in the cache from one
// kernel to the next and keep plain stores (128^3 lost 9 % with nt).  Plain on the host.
FG_HD void cstore_stream(cplx* p, cplx v, int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (nt) {
    typedef double fg_v2d __attribute__((ext_vector_type(2)));
    fg_v2d t;
    t.x = v.re;
    t.y = v.im;
    __builtin_nontemporal_store(t, reinterpret_cast<fg_v2d*>(p));
    return;
  }
#endif
  (void)nt;
  *p = v;
}

// The matching load (nt & 2): a line an FFT pass reads once.
FG_HD cplx cload_stream(const cplx* p, int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (nt & 2) {
    typedef double fg_v2d __attribute__((ext_vector_type(2)));
    const fg_v2d t = __builtin_nontemporal_load(reinterpret_cast<const fg_v2d*>(p));
    return cmake(t.x, t.y);
  }
#endif
  (void)nt;
  return *p;
}

// Geometry of one padded field component: the reference's in-place r2c layout,
// real [nx][ny][nzp], z fastest, complex view [nx][ny][nzc] on the same bytes -- except that the
// row pitch is rounded up to 128 bytes (8 complex) so that every row, FFT tile and halo plane
// starts on a cache-line boundary (nz/2+1 = 257 complex = 4112 B rows made every strided FFT pass
// fetch 1.5x its bytes).  nzf = nz/2+1 is the number of frequencies actually stored per row.
struct Grid {
  int nx, ny, nz;
  int nzf;      // nz/2+1 complex coefficients per row
  int nzc;      // complex row pitch >= nzf (multiple of 8 for nz >= 64)
  int nzp;      // real row pitch = 2*nzc
  long nyzp;    // ny*nzp
  long n;       // nx*ny*nzp padded reals per component
  long nxyz;    // nx*ny*nz
  double dx, dy, dz;
  double hx, hy, hz;  // voxels per unit length of the GLOBAL grid: n_a / d_a  (F:18618-18620)
  // x neighbours of the marching displacement sweep: plane q < 0 lives at q + xw_lo, plane q >= nx at q - xw_hi.
  // Periodic field: both nx.  x-slab of a decomposed grid (the component carries 4 spare planes behind its nx own
  // ones): plane nx (first plane of the right neighbour) at nx, plane -1 (last plane of the left neighbour) at nx + 3.
  int xw_lo, xw_hi;
};

FG_HD Grid make_grid(int nx, int ny, int nz, double dx, double dy, double dz) {
  Grid g;
  g.nx = nx; g.ny = ny; g.nz = nz;
  g.nzf = nz / 2 + 1;
  g.nzc = (nz >= 64) ? ((g.nzf + 7) / 8) * 8 : g.nzf;
  g.nzp = 2 * g.nzc;
  g.nyzp = (long)ny * g.nzp;
  g.n = (long)nx * g.nyzp;
  g.nxyz = (long)nx * ny * nz;
  g.dx = dx; g.dy = dy; g.dz = dz;
  g.hx = nx / dx; g.hy = ny / dy; g.hz = nz / dz;
  g.xw_lo = g.xw_hi = nx;
  return g;
}

}  // namespace fg
