// 3-D real-to-complex / complex-to-real FFT plan on one MI355X (host interface).
//
// Same contract as the reference's FFT3<double> (F:7203-7245, FFTW r2c/c2r):
// in place on a padded component [nx][ny][nzp] <-> [nx][ny][nzc] complex,
// unnormalised, forward e^{-i}; a scale factor can be folded into the last
// forward pass (the reference multiplies by 1/N in a separate sweep, F:18501-18506).
#pragma once

#include <hip/hip_runtime.h>

#include "fg_common.h"
#include "fg_fft_smooth.h"

namespace fg {

// Green-operator inputs of the fused pass: separable k tables (kpm[a], kp[a]; for the axis that is
// transformed the table index is the frequency, tables 1 and 2 are indexed by ky / kz) and c10, c20.
struct G0Params {
  const double* kpm[3];
  const cplx* kp[3];
  double c10, c20;
  double inv_h0;   // 2 n / d of the axis the fused pass transforms (physical x)
};

// A run of x planes [x0, x0 + np) of the field (np < 0: all).  The z and y passes are local to x planes: the solver runs
// them chunk by chunk (r2c(c) -> y(c), y^-1(c) -> c2r(c)) so that the hand-over between the two stays in the 256 MB Infinity
// Cache (Solver::fft_g0_chain, option pair_chunk).  Same kernels, same per-line arithmetic: bit-identical to the whole-field
// passes.  nt >= 0 overrides the streaming-access flags of the launch (bit 0 stores, bit 1 loads).  Power-of-two lengths only.
struct PlaneWindow {
  int x0 = 0;
  int np = -1;
  int nt = -1;
};

class Fft3 {
 public:
  Fft3(const Grid& g, hipStream_t stream);
  ~Fft3();
  Fft3(const Fft3&) = delete;
  Fft3& operator=(const Fft3&) = delete;

  // data: ncomp components, comp_stride doubles apart, each g.n doubles.
  void forward(double* data, int ncomp, long comp_stride, double scale);
  void inverse(double* data, int ncomp, long comp_stride);

  // single-axis entry points (used by the slab-decomposed driver and the stage tests)
  void r2c_z(double* data, int ncomp, long comp_stride, const PlaneWindow* w = nullptr);
  void c2c_y(double* data, int ncomp, long comp_stride, int dir, double scale, const PlaneWindow* w = nullptr);
  void c2c_x(double* data, int ncomp, long comp_stride, int dir, double scale);
  void c2r_z(double* data, int ncomp, long comp_stride, const PlaneWindow* w = nullptr);
  bool can_window() const { return fast_[1] && fast_[2]; }   // the passes that take a PlaneWindow
  void scale(double* data, int ncomp, long comp_stride, double scale);
  // Slab decomposition (x-slab side): the y pass with the all-to-all layout on one side.  Plain layout
  // [nx][ny][nzc]; blocked layout [q][nx][ny/P][nzc] (block q = what peer q receives / sent), see StridedArgs.
  // dir = -1: in plain -> out blocked (forward); dir = +1: in blocked -> out plain (inverse).  Component strides in doubles.
  bool can_block_y(int nranks) const;
  void c2c_y_blocked(double* in, long in_cs, double* out, long out_cs, int ncomp, int dir, double scale, int nranks,
                     int interleave = 1);
  // x-contiguous intermediate layout between the y passes and the fused x pass: [zc/8][y][x][8] complex, so that a tile of the
  // fused pass (8 kz columns x all x of one ky) is ONE contiguous run of nx * 128 bytes instead of nx segments ny*nzc*16 B
  // apart; the scatter moves to the y passes (dir = -1: in plain -> out x-layout; dir = +1: in x-layout -> out plain),
  // fused_g0(..., xlayout = true) works on it in place.  Out of place (in != out).
  bool can_xlayout() const;
  void c2c_y_xlayout(double* in, long in_cs, double* out, long out_cs, int ncomp, int dir, double scale,
                     const PlaneWindow* w = nullptr);
  // small grids: the z and y transforms of a z-y plane in one kernel (the complex plane [ny][nzc] in LDS, fg_fft_plane.h);
  // dir = -1: r2c along z then c2c along y (= r2c_z + c2c_y), dir = +1: the inverse pair (= c2c_y + c2r_z), in place
  bool can_plane() const;
  void zy_plane(double* data, int ncomp, long comp_stride, int dir);
  bool can_fuse(int axis, int ncomp = 3) const;
  // ncomp = 3: elastic Green operator on three components; ncomp = 1: scalar (heat / porous) operator c10 / |k|^2
  // xjump != 0 (doubles): line point j of the x pass sits at j * ls + (j >> xsplit) * xjump (interleaved slab layout)
  void fused_g0(double* data, long comp_stride, int axis, double scale, const G0Params& gp, int jj0, int ncomp = 3, int xsplit = 31,
                long xjump = 0, bool xlayout = false);

  // tile kernels' fused x pass of three components on one joint image (fg_fft_smooth.h SmoothPlan::joint; default) or on one
  // image per component: the same butterflies in either form
  void set_joint_x(bool on);

  bool fast_x() const { return fast_[0]; }
  bool fast_y() const { return fast_[1]; }
  bool fast_z() const { return fast_[2]; }
  const cplx* x_twiddles() const { return tw_[0]; }
  const cplx* z_twiddles() const { return tw_[2]; }  // pass twiddles of M = nz/2 (fast z path)
  const cplx* z_roots() const { return wz_; }        // e^{-2 pi i k/nz}, k = 0..nz/2

 private:
  void strided(double* data, int ncomp, long comp_stride, int axis, int dir, double scale, const PlaneWindow* w = nullptr);
  Grid g_;
  hipStream_t stream_;
  bool fast_[3];
  int odd_[3];       // odd p <= 25: the axis length (z: nz/2) is that factor times a fast power of two; 0 otherwise
  fft::SmoothPlan smooth_[3];   // n != 0: the axis (z: nz/2, odd nz: nz) runs the Stockham tile kernels of fg_fft_smooth.h
  bool zodd_ = false;           // odd nz with a plan: the rows are transformed as nz complex points
  fft::SmoothPlan xfused_plan_[2];   // fused x pass of the tile kernels: [0] one component, [1] three (n = 0: none)
  bool joint_x_ = true;
  int stream_stores_ = 0;  // FFT passes use cache-bypassing stores (fields larger than the Infinity Cache)
  cplx* tw_[3];      // per-axis pass twiddles (fast path) ; z: for M = nz/2
  cplx* half_root_[2];  // e^{-i pi j/n}, j < n/8, of x and y (fused Green-operator pass)
  cplx* wz_;         // w^k = e^{-2 pi i k/nz}, k = 0..nz/2   (fast z path)
  cplx* wgen_[3];    // e^{-2 pi i k/n}, k = 0..n-1           (generic path)
  double* scratch_;  // one padded component (generic path)
};

}  // namespace fg
