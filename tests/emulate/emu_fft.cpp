// Host emulation of the FFT kernels' thread bodies (test infrastructure).
// Runs every phase for all threads of a block in turn, which is what the
// device does between __syncthreads(); build with g++ -DFG_HOST_EMULATION.
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "../../fibergen_amd/csrc/fg_fft_kernels.h"
#include "../../fibergen_amd/csrc/fg_fft_plane.h"
#include "../../fibergen_amd/csrc/fg_fft_smooth.h"
#include "../../fibergen_amd/csrc/fg_fft_smooth_plans.h"
#include "../../fibergen_amd/csrc/fg_fft_tables.h"

using namespace fg;
using namespace fg::fft;

template <class K, class Args, int PH>
struct PhaseLoop {
  static void run(std::vector<typename K::Regs>& regs, int block, double* lds, const Args& a) {
    for (int tid = 0; tid < K::THREADS; ++tid) K::template phase<PH>(regs[tid], block, tid, lds, a);
    if constexpr (PH + 1 < K::NPHASE) PhaseLoop<K, Args, PH + 1>::run(regs, block, lds, a);
  }
};

// Kernels that replace some workgroup barriers by wave-local fences (K::barrier_after(PH): 2 = __syncthreads, 1 = fence of
// one wave, see fg_fft_plane.h) are emulated accordingly: the phases between two workgroup barriers run WAVE BY WAVE -- all
// phases of that stretch for the 64 threads of one wave, then the next wave -- in an order that alternates from block to
// block (highest wave first / lowest wave first).  A dependency between waves that sits behind a mere wave fence then reads
// LDS the other wave has not written yet (NaN or a stale value) in one of the two orders, and the comparison with numpy fails.
template <class K, class = void>
struct HasFenceTable : std::false_type {};
template <class K>
struct HasFenceTable<K, std::void_t<decltype(K::barrier_after(0))>> : std::bool_constant<!std::is_same_v<decltype(K::barrier_after(0)), bool>> {};

template <class K, class Args, int PH>
struct FencedLoop {
  // runs phases PH .. (last phase of the stretch that starts at PH) for threads [t0, t1); returns nothing: the stretch end is static
  static constexpr int stretch_end(int ph) { return (ph + 1 >= K::NPHASE || K::barrier_after(ph) == 2) ? ph : stretch_end(ph + 1); }
  template <int P, int END>
  static void wave(std::vector<typename K::Regs>& regs, int block, double* lds, const Args& a, int t0, int t1) {
    for (int tid = t0; tid < t1; ++tid) K::template phase<P>(regs[tid], block, tid, lds, a);
    if constexpr (P < END) wave<P + 1, END>(regs, block, lds, a, t0, t1);
  }
  static void run(std::vector<typename K::Regs>& regs, int block, double* lds, const Args& a) {
    constexpr int END = stretch_end(PH);
    constexpr int NW = (K::THREADS + 63) / 64;
    for (int i = 0; i < NW; ++i) {
      const int w = (block & 1) ? i : NW - 1 - i;
      const int t0 = w * 64, t1 = t0 + 64 < K::THREADS ? t0 + 64 : K::THREADS;
      wave<PH, END>(regs, block, lds, a, t0, t1);
    }
    if constexpr (END + 1 < K::NPHASE) FencedLoop<K, Args, END + 1>::run(regs, block, lds, a);
  }
};

template <class K, class Args>
static void run_blocks(long nblocks, const Args& a) {
  std::vector<typename K::Regs> regs(K::THREADS);
  std::vector<double> lds(K::LDS_DOUBLES);
  for (long b = 0; b < nblocks; ++b) {
    for (auto& x : lds) x = NAN;  // catch reads of never-written LDS
    if constexpr (HasFenceTable<K>::value) FencedLoop<K, Args, 0>::run(regs, (int)b, lds.data(), a);
    else PhaseLoop<K, Args, 0>::run(regs, (int)b, lds.data(), a);
  }
}

template <int N, int C>
static void strided_dir(StridedArgs a, long nblocks, int dir) {
  if (dir < 0) run_blocks<StridedKernel<N, C, -1>, StridedArgs>(nblocks, a);
  else run_blocks<StridedKernel<N, C, +1>, StridedArgs>(nblocks, a);
}

// Stockham tile kernels (fg_fft_smooth.h): the phases of k_smooth_strided / k_smooth_z for every thread of every workgroup,
// in the device's order (load | barrier | per pass: read + butterfly, barrier, write, barrier | store), with the plan the
// library's planner makes.  Returns 1 when the length has no plan (a prime factor above 13).
template <int DIR, bool FDIV = true>
static void emu_smooth_passes(cplx* img, const SmoothPlan& plan, const SmoothMap& L, const cplx* w, int wscale) {
  constexpr int QMAX = 8;   // a thread owns up to smooth_rounds(R) butterflies: virtual threads tid + q * threads
  std::vector<cplx> regs((size_t)plan.threads * QMAX * kSmoothMaxRadix);
  std::vector<char> active((size_t)plan.threads * QMAX);
  int Ns = 1;
  for (int f = 0; f < plan.npass; ++f) {
    const int R = plan.fac[f];
    const int nvirt = plan.threads * (plan.cap ? smooth_rounds(R, plan.cap) : 1);
    for (int half = 0; half < 2; ++half)
      for (int tid = 0; tid < nvirt; ++tid) {
        cplx* v = &regs[(size_t)tid * kSmoothMaxRadix];
        switch (R) {
#define FG_R(r)                                                                                              \
  case r:                                                                                                    \
    if (half == 0) active[tid] = smooth_pass_read<r, DIR, FDIV>(img, plan.n, Ns, L, w, wscale, tid, v);            \
    else if (active[tid]) smooth_pass_write<r, FDIV>(img, plan.n, Ns, L, tid, v);                                  \
    break;
          FG_R(2) FG_R(3) FG_R(4) FG_R(5) FG_R(6) FG_R(7) FG_R(8) FG_R(9) FG_R(10) FG_R(11) FG_R(12) FG_R(13) FG_R(14) FG_R(15)
          FG_R(16) FG_R(18) FG_R(20) FG_R(21) FG_R(22) FG_R(24) FG_R(25) FG_R(26) FG_R(27) FG_R(28) FG_R(30) FG_R(32)
#undef FG_R
          default: std::abort();
        }
      }
    Ns *= R;
  }
}

extern "C" {

// c2c along a strided axis of data[nouter][N][ncols] (ls = ncols, os = N*ncols), device tile geometry
int emu_strided(int N, int dir, double* data, int ncols, int nouter, double scale) {
  std::vector<cplx> tw = make_pass_twiddles(N);
  StridedArgs a;
  a.nt = 0;
  a.xcd_order = 0;
  a.data = reinterpret_cast<cplx*>(data);
  a.ls = ncols;
  a.os = (long)N * ncols;
  a.ncols = ncols;
  a.scale = scale;
  a.tw = tw.data();
#define CASE(n)                                                  \
  if (N == n) {                                                  \
    constexpr int C = TileCols<n>::value;                        \
    a.tiles_per_outer = (ncols + C - 1) / C;                     \
    strided_dir<n, C>(a, (long)a.tiles_per_outer * nouter, dir); \
    return 0;                                                    \
  }
  CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) CASE(512) CASE(1024)
#undef CASE
  return 1;
}

int emu_r2c(int nz, double* data, long nrows) {
  int M = nz / 2;
  std::vector<cplx> tw = make_pass_twiddles(M);
  std::vector<cplx> wz = make_unit_roots(nz, M + 1);
  ZArgs a = {data, nrows, 2 * (nz / 2 + 1), tw.data(), wz.data()};
#define CASE(m)                                                              \
  if (M == m) {                                                              \
    constexpr int l = ZLines<m>::value;                                      \
    run_blocks<R2CKernel<m, l>, ZArgs>((nrows + l - 1) / l, a);              \
    return 0;                                                                \
  }
  CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) CASE(512) CASE(1024)
#undef CASE
  return 1;
}

int emu_c2r(int nz, double* data, long nrows) {
  int M = nz / 2;
  std::vector<cplx> tw = make_pass_twiddles(M);
  std::vector<cplx> wz = make_unit_roots(nz, M + 1);
  ZArgs a = {data, nrows, 2 * (nz / 2 + 1), tw.data(), wz.data()};
#define CASE(m)                                                              \
  if (M == m) {                                                              \
    constexpr int l = ZLines<m>::value;                                      \
    run_blocks<C2RKernel<m, l>, ZArgs>((nrows + l - 1) / l, a);              \
    return 0;                                                                \
  }
  CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) CASE(512) CASE(1024)
#undef CASE
  return 1;
}
// fused x pass: data[3][N][ncols] complex (ls = ncols, comp stride N*ncols), columns flattened (ky,kz)
int emu_xfused(int N, double* data, int ny, int nzc, int nzf, double scale, double c10, double c20,
               const double* kpm0, const double* kp0, const double* kpm1, const double* kp1, const double* kpm2,
               const double* kp2) {
  std::vector<cplx> tw = make_pass_twiddles(N);
  XFusedArgs a;
  a.nt = 0;
  a.xcd_order = 0;
  a.data = reinterpret_cast<cplx*>(data);
  a.ncols = ny * nzc;
  a.comp_stride = (long)N * a.ncols;
  a.ls = a.ncols;
  a.os = 0;
  a.flat_cols = 1;
  a.nzc = nzc;
  a.nzf = nzf;
  a.jj0 = 0;
  a.scale = scale;
  a.c10 = c10;
  a.c20 = c20;
  a.tw = tw.data();
  // the transformed axis' factors are rebuilt from theta = pi kx / N: e^{-i pi j/N} and 1/h read off the table (kpm0[1] = sin(pi/N)/h)
  std::vector<cplx> half_root = make_unit_roots(2 * N, N / 8 > 0 ? N / 8 : 1);
  a.half_root = half_root.data();
  a.inv_h = N > 1 ? kpm0[1] / std::sin(3.14159265358979323846 / N) : 1.0;
  for (int q = 0; q < 8; ++q) a.xq[q] = cmake(1.0, 0.0);   // filled per N below
  a.kpm[0] = kpm0; a.kpm[1] = kpm1; a.kpm[2] = kpm2;
  a.kp[0] = reinterpret_cast<const cplx*>(kp0);
  a.kp[1] = reinterpret_cast<const cplx*>(kp1);
  a.kp[2] = reinterpret_cast<const cplx*>(kp2);
#define CASE(n)                                                               \
  if (N == n) {                                                               \
    constexpr int C = n == 1024 ? 4 : 8; /* the product's tile width for three components */ \
    a.tiles_per_outer = (a.ncols + C - 1) / C;                                \
    for (int q = 0; q < 8; ++q) {                                             \
      const double th = 3.14159265358979323846 * fft::Line<n>::last_index(0, q) / n;      \
      a.xq[q] = cmake(std::cos(th), std::sin(th));                            \
    }                                                                         \
    run_blocks<XFusedKernel<n, C>, XFusedArgs>((long)a.tiles_per_outer, a);   \
    return 0;                                                                 \
  }
  CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) CASE(512) CASE(1024)
#undef CASE
  return 1;
}

// The y pass with the x-contiguous layout [zc/8][y][x][8] on one side (Fft3::c2c_y_xlayout's strides, 8-column tiles):
// dir < 0: in plain [nx][NY][nzc] -> out x-layout; dir > 0: in x-layout -> out plain
int emu_strided_xlayout(int NY, int dir, double* in, double* out, int nx, int nzc, double scale) {
  std::vector<cplx> tw = make_pass_twiddles(NY);
  StridedArgs a;
  a.nt = 0;
  a.xcd_order = 0;
  a.data = reinterpret_cast<cplx*>(in);
  a.out = reinterpret_cast<cplx*>(out);
  a.out_cs = 0;
  a.ncols = nzc;
  a.scale = scale;
  a.tw = tw.data();
  a.tiles_per_outer = (nzc + 7) / 8;
  const long plain_ls = nzc, plain_os = (long)NY * nzc, xl_ls = (long)nx * 8, xl_os = 8, xl_ts = (long)NY * nx * 8;
  if (dir < 0) {
    a.ls = plain_ls; a.os = plain_os; a.ls_out = xl_ls; a.os_out = xl_os; a.ts_out = xl_ts;
  } else {
    a.ls = xl_ls; a.os = xl_os; a.ts_in = xl_ts; a.ls_out = plain_ls; a.os_out = plain_os;
  }
#define CASE(n)                                                      \
  if (NY == n) {                                                     \
    strided_dir<n, 8>(a, (long)a.tiles_per_outer * nx, dir);         \
    return 0;                                                        \
  }
  CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) CASE(512)
#undef CASE
  return 1;
}

// the fused x pass on the x-contiguous layout: data[3][nzc/8][ny][N][8]
int emu_xfused_xlayout(int N, double* data, int ny, int nzc, int nzf, double scale, double c10, double c20,
                       const double* kpm0, const double* kp0, const double* kpm1, const double* kp1, const double* kpm2,
                       const double* kp2) {
  std::vector<cplx> tw = make_pass_twiddles(N);
  XFusedArgs a;
  a.nt = 0;
  a.xcd_order = 0;
  a.data = reinterpret_cast<cplx*>(data);
  a.ncols = ny * nzc;
  a.comp_stride = (long)N * a.ncols;
  a.ls = 8;
  a.os = 0;
  a.flat_cols = 0;
  a.xl_ny = ny;
  a.nzc = nzc;
  a.nzf = nzf;
  a.jj0 = 0;
  a.scale = scale;
  a.c10 = c10;
  a.c20 = c20;
  a.tw = tw.data();
  std::vector<cplx> half_root = make_unit_roots(2 * N, N / 8 > 0 ? N / 8 : 1);
  a.half_root = half_root.data();
  a.inv_h = N > 1 ? kpm0[1] / std::sin(3.14159265358979323846 / N) : 1.0;
  a.kpm[0] = kpm0; a.kpm[1] = kpm1; a.kpm[2] = kpm2;
  a.kp[0] = reinterpret_cast<const cplx*>(kp0);
  a.kp[1] = reinterpret_cast<const cplx*>(kp1);
  a.kp[2] = reinterpret_cast<const cplx*>(kp2);
#define CASE(n)                                                               \
  if (N == n) {                                                               \
    a.tiles_per_outer = a.ncols / 8;                                          \
    for (int q = 0; q < 8; ++q) {                                             \
      const double th = 3.14159265358979323846 * fft::Line<n>::last_index(0, q) / n;      \
      a.xq[q] = cmake(std::cos(th), std::sin(th));                            \
    }                                                                         \
    run_blocks<XFusedKernel<n, 8>, XFusedArgs>((long)a.tiles_per_outer, a);   \
    return 0;                                                                 \
  }
  CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) CASE(512)
#undef CASE
  return 1;
}

// generic fall-backs
void emu_dft_strided(const double* src, double* dst, int n, int ncols, int nouter, int dir, double scale) {
  std::vector<cplx> w = make_unit_roots(n, n);
  for (int o = 0; o < nouter; ++o)
    for (int c = 0; c < ncols; ++c)
      for (int k = 0; k < n; ++k)
        dft_strided_point(reinterpret_cast<const cplx*>(src), reinterpret_cast<cplx*>(dst),
                          (long)o * n * ncols + c, ncols, n, k, dir, scale, w.data());
}
void emu_r2c_generic(const double* src, double* dst, int nz, long nrows) {
  std::vector<cplx> w = make_unit_roots(nz, nz);
  int nzc = nz / 2 + 1, nzp = 2 * nzc;
  for (long r = 0; r < nrows; ++r)
    for (int k = 0; k < nzc; ++k) r2c_point(src + r * nzp, reinterpret_cast<cplx*>(dst + r * nzp), nz, k, w.data());
}
void emu_c2r_generic(const double* src, double* dst, int nz, long nrows) {
  std::vector<cplx> w = make_unit_roots(nz, nz);
  int nzc = nz / 2 + 1, nzp = 2 * nzc;
  for (long r = 0; r < nrows; ++r)
    for (int m = 0; m < nz; ++m) c2r_point(reinterpret_cast<const cplx*>(src + r * nzp), dst + r * nzp, nz, m, w.data());
}

// z + y transforms of whole planes in one kernel (fg_fft_plane.h): data[nplanes][ny][2*nzc] real rows, in place
int emu_plane(int ny, int nz, int dir, double* data, int nplanes) {
  const int M = nz / 2, nzc = nz / 2 + 1;
  std::vector<cplx> twz = make_pass_twiddles(M), twy = make_pass_twiddles(ny);
  std::vector<cplx> wz = make_unit_roots(nz, M + 1);
  PlaneArgs a = {data, (long)ny * 2 * nzc, 2 * nzc, twz.data(), wz.data(), twy.data()};
#define CASE(n, m)                                                        \
  if (ny == n && M == m) {                                                \
    if (dir < 0) run_blocks<ZYKernel<n, m>, PlaneArgs>(nplanes, a);       \
    else run_blocks<YZKernel<n, m>, PlaneArgs>(nplanes, a);               \
    return 0;                                                             \
  }
  CASE(16, 8) CASE(16, 16) CASE(16, 32) CASE(16, 64) CASE(32, 8) CASE(32, 16) CASE(32, 32) CASE(32, 64) CASE(64, 8) CASE(64, 16)
  CASE(64, 32) CASE(64, 64) CASE(128, 8) CASE(128, 16) CASE(128, 32) CASE(128, 64) CASE(256, 8) CASE(256, 16) CASE(256, 32)
#undef CASE
  return 1;
}

int emu_smooth_strided(int N, int dir, double* data, int ncols, int nouter, double scale, int* plan_out) {
  SmoothArgs a;
  if (!smooth_plan_strided(N, &a.plan)) return 1;
  if (plan_out) {
    plan_out[0] = a.plan.lines, plan_out[1] = a.plan.threads, plan_out[2] = a.plan.npass;
    for (int i = 0; i < a.plan.npass; ++i) plan_out[3 + i] = a.plan.fac[i];
  }
  std::vector<cplx> w = make_unit_roots(N, N);
  const int C = a.plan.lines, T = a.plan.threads;
  a.data = reinterpret_cast<cplx*>(data);
  a.ls = ncols;
  a.os = (long)N * ncols;
  a.ncols = ncols;
  a.tiles_per_outer = (ncols + C - 1) / C;
  a.scale = scale;
  a.w = w.data();
  a.nt = 0;
  std::vector<cplx> img((size_t)N * C);
  const SmoothMap L = {C, 1, C, false};
  for (int b = 0; b < a.tiles_per_outer * nouter; ++b) {
    for (auto& x : img) x = cmake(NAN, NAN);
    for (int tid = 0; tid < T; ++tid) {   // (the emulation takes the widest variant: more empty slots, the same elements)
      if (C == 8) smooth_strided_load<8, 16>(a, b, tid, T, img.data());
      else if (C == 16) smooth_strided_load<16, 16>(a, b, tid, T, img.data());
      else if (C == 32) smooth_strided_load<32, 16>(a, b, tid, T, img.data());
      else if (C == 4) smooth_strided_load<4, 16>(a, b, tid, T, img.data());
      else smooth_strided_load<2, 16>(a, b, tid, T, img.data());
    }
    if (dir < 0) emu_smooth_passes<-1>(img.data(), a.plan, L, a.w, 1);
    else emu_smooth_passes<+1>(img.data(), a.plan, L, a.w, 1);
    for (int tid = 0; tid < T; ++tid) {
      if (C == 8) smooth_strided_store<8>(a, b, tid, T, img.data());
      else if (C == 16) smooth_strided_store<16>(a, b, tid, T, img.data());
      else if (C == 32) smooth_strided_store<32>(a, b, tid, T, img.data());
      else if (C == 4) smooth_strided_store<4>(a, b, tid, T, img.data());
      else smooth_strided_store<2>(a, b, tid, T, img.data());
    }
  }
  return 0;
}

// fused x pass of the tile kernels (k_smooth_xfused / k_smooth_xjoint): data[3][N][ncols], columns flattened (ky, kz);
// joint = 1: the three components on one joint image (what the planner picks where it takes no more passes), 0: one image each
int emu_smooth_xfused(int N, double* data, int ny, int nzc, int nzf, double scale, double c10, double c20, const double* kpm0,
                      const double* kp0, const double* kpm1, const double* kp1, const double* kpm2, const double* kp2, int joint,
                      int* plan_out) {
  SmoothXArgs a;
  if (!smooth_plan_xfused(N, 3, &a.base.plan, joint != 0)) return 1;
  const SmoothPlan& plan = a.base.plan;
  if (plan_out) {
    plan_out[0] = plan.lines, plan_out[1] = plan.threads, plan_out[2] = plan.npass, plan_out[3] = plan.joint, plan_out[4] = plan.cap;
    for (int i = 0; i < plan.npass; ++i) plan_out[5 + i] = plan.fac[i];
  }
  std::vector<cplx> w = make_unit_roots(N, N);
  const bool jn = plan.joint >= 1;
  const int C = jn ? plan.lines / plan.joint : plan.lines, T = plan.threads;
  a.base.data = reinterpret_cast<cplx*>(data);
  a.base.ncols = ny * nzc;
  a.base.ls = a.base.ncols;
  a.base.os = 0;
  a.base.tiles_per_outer = (a.base.ncols + C - 1) / C;
  a.base.scale = scale;
  a.base.w = w.data();
  a.base.nt = 0;
  a.comp_stride = (long)N * a.base.ncols;
  a.ncomp = 3;
  a.nzc = nzc, a.nzf = nzf, a.jj0 = 0;
  a.c10 = c10, a.c20 = c20;
  a.kpm[0] = kpm0, a.kpm[1] = kpm1, a.kpm[2] = kpm2;
  a.kp[0] = reinterpret_cast<const cplx*>(kp0), a.kp[1] = reinterpret_cast<const cplx*>(kp1), a.kp[2] = reinterpret_cast<const cplx*>(kp2);
  std::vector<cplx> img((size_t)3 * N * C);
  const long comp = (long)N * C;
  for (int b = 0; b < a.base.tiles_per_outer; ++b) {
    for (auto& x : img) x = cmake(NAN, NAN);
    if (jn) {
      if (C != 4 && C != 8 && C != 16) return 2;
      const SmoothMap L = {3 * C, 1, 3 * C, false};
      for (int tid = 0; tid < T; ++tid) {
        if (C == 4) smooth_joint_load<4, 3, 16>(a, b, tid, T, img.data());
        else if (C == 8) smooth_joint_load<8, 3, 16>(a, b, tid, T, img.data());
        else smooth_joint_load<16, 3, 16>(a, b, tid, T, img.data());
      }
      emu_smooth_passes<-1, false>(img.data(), plan, L, a.base.w, 1);
      for (int tid = 0; tid < T; ++tid) {
        if (C == 4) smooth_joint_green<4, 3>(a, b, tid, T, img.data());
        else if (C == 8) smooth_joint_green<8, 3>(a, b, tid, T, img.data());
        else smooth_joint_green<16, 3>(a, b, tid, T, img.data());
      }
      emu_smooth_passes<+1, false>(img.data(), plan, L, a.base.w, 1);
      for (int tid = 0; tid < T; ++tid) {
        if (C == 4) smooth_joint_store<4, 3>(a, b, tid, T, img.data());
        else if (C == 8) smooth_joint_store<8, 3>(a, b, tid, T, img.data());
        else smooth_joint_store<16, 3>(a, b, tid, T, img.data());
      }
      continue;
    }
    const SmoothMap L = {C, 1, C, false};
    for (int c = 0; c < 3; ++c) {
      SmoothArgs ac = a.base;
      ac.data += c * a.comp_stride;
      for (int tid = 0; tid < T; ++tid) {
        if (C == 8) smooth_strided_load<8, 16>(ac, b, tid, T, img.data() + c * comp);
        else if (C == 16) smooth_strided_load<16, 16>(ac, b, tid, T, img.data() + c * comp);
        else smooth_strided_load<4, 16>(ac, b, tid, T, img.data() + c * comp);
      }
    }
    for (int c = 0; c < 3; ++c) emu_smooth_passes<-1, false>(img.data() + c * comp, plan, L, a.base.w, 1);
    for (int tid = 0; tid < T; ++tid) {
      if (C == 8) smooth_x_green<8, 3>(a, b, tid, T, img.data());
      else if (C == 16) smooth_x_green<16, 3>(a, b, tid, T, img.data());
      else smooth_x_green<4, 3>(a, b, tid, T, img.data());
    }
    for (int c = 0; c < 3; ++c) emu_smooth_passes<+1, false>(img.data() + c * comp, plan, L, a.base.w, 1);
    for (int c = 0; c < 3; ++c) {
      SmoothArgs ac = a.base;
      ac.data += c * a.comp_stride;
      ac.scale = 1.0;
      for (int tid = 0; tid < T; ++tid) {
        if (C == 8) smooth_strided_store<8>(ac, b, tid, T, img.data() + c * comp);
        else if (C == 16) smooth_strided_store<16>(ac, b, tid, T, img.data() + c * comp);
        else smooth_strided_store<4>(ac, b, tid, T, img.data() + c * comp);
      }
    }
  }
  return 0;
}

// the plan kernels' tables (fg_fft_smooth_plans.h) against the planner: number of entries whose plan the planner does NOT make
// (such an entry would never be launched: its grid size runs the class kernels)
int emu_plan_table_mismatches(int* nentries) {
  int bad = 0, n = 0;
  SmoothPlan p;
#define FG_X(N, C, R0, R1, R2) ++n; if (!smooth_plan_strided(N, &p) || !smooth_plan_is(p, N, C, 256, 20, R0, R1, R2)) ++bad;
  FG_SMOOTH_STRIDED_PLANS(FG_X)
#undef FG_X
#define FG_X(M, LINES, R0, R1, R2) ++n; if (!smooth_plan_z(M, &p) || !smooth_plan_is(p, M, LINES, 256, 20, R0, R1, R2)) ++bad;
  FG_SMOOTH_Z_PLANS(FG_X)
#undef FG_X
#define FG_X(N, C, T, CAP, R0, R1, R2) ++n; if (!smooth_plan_xfused(N, 3, &p) || p.joint != 3 || !smooth_plan_is(p, N, 3 * C, T, CAP, R0, R1, R2)) ++bad;
  FG_SMOOTH_X_PLANS(FG_X)
#undef FG_X
#define FG_X(N, C, T, CAP, R0, R1, R2) ++n; if (!smooth_plan_xfused(N, 1, &p) || p.joint != 1 || !smooth_plan_is(p, N, C, T, CAP, R0, R1, R2)) ++bad;
  FG_SMOOTH_X1_PLANS(FG_X)
#undef FG_X
  if (nentries) *nentries = n;
  return bad;
}

int emu_smooth_z(int nz, int fwd, double* data, long nrows, int* plan_out) {
  SmoothZArgs a;
  if (nz % 2) {   // odd nz: the rows as nz complex points
    if (!smooth_plan_z(nz, &a.plan)) return 1;
    a.odd = 1;
    if (plan_out) {
      plan_out[0] = a.plan.lines, plan_out[1] = a.plan.threads, plan_out[2] = a.plan.npass;
      for (int i = 0; i < a.plan.npass; ++i) plan_out[3 + i] = a.plan.fac[i];
    }
    std::vector<cplx> w = make_unit_roots(nz, nz);
    a.data = data;
    a.nrows = nrows;
    a.nzp = 2 * (nz / 2 + 1);
    a.w = w.data();
    a.nt = 0;
    const int pitch = smooth_z_pitch(nz), lines = a.plan.lines, T = a.plan.threads;
    std::vector<cplx> img((size_t)lines * pitch);
    const SmoothMap L = smooth_z_map(a.plan.n, lines);
    for (long b = 0; b * lines < nrows; ++b) {
      for (auto& x : img) x = cmake(NAN, NAN);
      const long row0 = b * lines;
      if (fwd) {
        for (int tid = 0; tid < T; ++tid) smooth_zodd_load_real<16>(a, row0, tid, T, img.data());
        emu_smooth_passes<-1>(img.data(), a.plan, L, a.w, 1);
        for (int tid = 0; tid < T; ++tid) smooth_zodd_store_half(a, row0, tid, T, img.data());
      } else {
        for (int tid = 0; tid < T; ++tid) smooth_zodd_load_half<16>(a, row0, tid, T, img.data());
        emu_smooth_passes<+1>(img.data(), a.plan, L, a.w, 1);
        for (int tid = 0; tid < T; ++tid) smooth_zodd_store_real(a, row0, tid, T, img.data());
      }
    }
    return 0;
  }
  const int M = nz / 2;
  if (!smooth_plan_z(M, &a.plan)) return 1;
  if (plan_out) {
    plan_out[0] = a.plan.lines, plan_out[1] = a.plan.threads, plan_out[2] = a.plan.npass;
    for (int i = 0; i < a.plan.npass; ++i) plan_out[3 + i] = a.plan.fac[i];
  }
  std::vector<cplx> w = make_unit_roots(nz, nz);
  a.data = data;
  a.nrows = nrows;
  a.nzp = 2 * (nz / 2 + 1);
  a.w = w.data();
  a.nt = 0;
  const int pitch = smooth_z_pitch(M), lines = a.plan.lines, T = a.plan.threads;
  std::vector<cplx> img((size_t)lines * pitch);
  const SmoothMap L = smooth_z_map(a.plan.n, lines);
  for (long b = 0; b * lines < nrows; ++b) {
    for (auto& x : img) x = cmake(NAN, NAN);
    const long row0 = b * lines;
    if (fwd) {
      for (int tid = 0; tid < T; ++tid) smooth_z_load_packed<16>(a, row0, tid, T, img.data());
      emu_smooth_passes<-1>(img.data(), a.plan, L, a.w, 2);
      for (int tid = 0; tid < T; ++tid) smooth_z_split_store<16>(a, row0, tid, T, img.data());
    } else {
      for (int tid = 0; tid < T; ++tid) smooth_z_load_spectrum<16>(a, row0, tid, T, img.data());
      for (int tid = 0; tid < T; ++tid) smooth_z_merge<16>(a, tid, T, img.data());
      emu_smooth_passes<+1>(img.data(), a.plan, L, a.w, 2);
      for (int tid = 0; tid < T; ++tid) smooth_z_store_packed(a, row0, tid, T, img.data());
    }
  }
  return 0;
}
}
