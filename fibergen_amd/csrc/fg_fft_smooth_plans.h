// The tile kernels built for ONE plan each (fg_fft_smooth_plans_*.hip): line length, tile shape and radices as template
// parameters.  The class kernels of fg_fft_smooth_yz.hip / _x.hip take any plan, but their register allocation is that of the
// largest butterfly they hold (the `switch` over radices) and every index division and LDS stride is a run-time value: the
// fused x pass of 200 points takes 256 VGPRs + 26 spilled in its class kernel and 116 built for its plan (20 x 10): 154 -> 112 us.
// The tables list the plans the planner (fg_fft_smooth.h) makes for the grid sizes users pick; any other plan -- and these,
// with smooth_plan_kernels(false) -- runs the class kernels: same butterflies, same results
// (tests/test_fft_emulation.py::test_plan_kernel_tables_match_the_planner holds the tables against the planner).
#pragma once
#include "fg_fft_smooth.h"

// strided (y / x) passes: X(N, columns per tile, R0, R1, R2)   (R2 = 1: two passes); 256 threads, <= 20 values per thread
#define FG_SMOOTH_STRIDED_PLANS(X)                                                                                         \
  X(100, 32, 10, 10, 1) X(120, 32, 15, 8, 1) X(150, 16, 15, 10, 1) X(180, 16, 15, 12, 1) X(200, 16, 8, 5, 5) X(240, 16, 16, 15, 1) \
  X(250, 16, 10, 5, 5) X(300, 8, 10, 10, 3) X(360, 8, 9, 8, 5) X(400, 8, 10, 10, 4) X(480, 8, 10, 8, 6) X(500, 8, 10, 10, 5)

// z passes (packed real rows of nz = 2 M points): X(M, rows per tile, R0, R1, R2); 256 threads
#define FG_SMOOTH_Z_PLANS(X)                                                                                               \
  X(50, 64, 10, 5, 1) X(60, 64, 10, 6, 1) X(75, 32, 15, 5, 1) X(90, 32, 10, 9, 1) X(100, 32, 10, 10, 1) X(120, 32, 15, 8, 1) \
  X(125, 32, 5, 5, 5) X(150, 16, 15, 10, 1) X(180, 16, 15, 12, 1) X(200, 16, 8, 5, 5) X(240, 16, 16, 15, 1) X(250, 16, 10, 5, 5)

// fused x pass on the joint image of three components: X(N, columns per tile, threads, values per thread, R0, R1, R2)
#define FG_SMOOTH_X_PLANS(X)                                                                                               \
  X(100, 16, 256, 20, 10, 10, 1) X(120, 8, 256, 20, 12, 10, 1) X(150, 8, 256, 20, 15, 10, 1) X(180, 8, 256, 20, 18, 10, 1)          \
  X(200, 8, 256, 20, 20, 10, 1) X(240, 8, 512, 20, 16, 15, 1) X(250, 8, 256, 32, 25, 10, 1) X(300, 8, 512, 20, 20, 15, 1)           \
  X(360, 8, 512, 20, 20, 18, 1) X(400, 8, 512, 20, 20, 20, 1) X(480, 4, 512, 20, 10, 8, 6) X(500, 4, 512, 20, 10, 10, 5)

namespace fg {
namespace fft {

// does `p` (lines = columns / rows per tile; the fused pass: joint * columns) equal the table entry?
inline bool smooth_plan_is(const SmoothPlan& p, int n, int lines, int threads, int cap, int r0, int r1, int r2) {
  return p.n == n && p.lines == lines && p.threads == threads && p.cap == cap && p.npass == (r2 > 1 ? 3 : 2) && p.fac[0] == r0 &&
         p.fac[1] == r1 && (r2 == 1 || p.fac[2] == r2);
}

// plan kernels on / off (process-wide; on by default)
void smooth_plan_kernels(bool on);
bool smooth_plan_kernels_on();

}  // namespace fft
}  // namespace fg
