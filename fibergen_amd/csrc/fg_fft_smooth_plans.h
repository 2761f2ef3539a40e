// The tile kernels built for ONE plan each (fg_fft_smooth_plans_*.hip): line length, tile shape and radices as template
// parameters.  The class kernels of fg_fft_smooth_yz.hip / _x.hip take any plan, but their register allocation is that of the
// largest butterfly they hold (the `switch` over radices) and every index division and LDS stride is a run-time value: the
// fused x pass of 200 points takes 256 VGPRs + 26 spilled in its class kernel and 116 built for its plan (20 x 10): 154 -> 112 us.
// The tables list the plans the planner (fg_fft_smooth.h) makes for 50 ... 1000 points in steps users pick (multiples of 10 with
// prime factors <= 13, and the z passes' halves of those); generated with the planner (tests/emulate: emu_smooth_* plan_out); any other plan -- and these,
// with smooth_plan_kernels(false) -- runs the class kernels: same butterflies, same results
// (tests/test_fft_emulation.py::test_plan_kernel_tables_match_the_planner holds the tables against the planner).
#pragma once
#include "fg_fft_smooth.h"

// strided (y / x) passes: X(N, columns per tile, R0, R1, R2)   (R2 = 1: two passes); 256 threads, several butterflies of a radix <= 10 per thread (<= 20 values)
#define FG_SMOOTH_STRIDED_PLANS(X) \
  /* p * 2^k lengths (run the tile kernels only where Fft3 prefers them to the sub-line kernels) */ \
  X(48, 32, 8, 6, 1) X(72, 32, 9, 8, 1) X(80, 32, 10, 8, 1) X(96, 32, 12, 8, 1) X(112, 32, 14, 8, 1) X(144, 16, 12, 12, 1) X(160, 16, 16, 10, 1) X(192, 16, 16, 12, 1) X(224, 16, 16, 14, 1) X(288, 8, 8, 6, 6) X(320, 8, 8, 8, 5) X(384, 8, 8, 8, 6) X(448, 8, 8, 8, 7) X(768, 8, 32, 24, 1) X(896, 8, 32, 28, 1) \
  X(50, 32, 10, 5, 1) X(60, 32, 10, 6, 1) X(70, 32, 10, 7, 1) X(90, 32, 10, 9, 1) X(100, 32, 10, 10, 1)                \
  X(110, 16, 11, 10, 1) X(120, 32, 15, 8, 1) X(130, 16, 13, 10, 1) X(140, 16, 14, 10, 1) X(150, 16, 15, 10, 1)         \
  X(180, 16, 15, 12, 1) X(200, 16, 8, 5, 5) X(210, 16, 15, 14, 1) X(220, 8, 11, 10, 2) X(240, 16, 16, 15, 1)           \
  X(250, 16, 10, 5, 5) X(260, 8, 13, 10, 2) X(270, 8, 9, 6, 5) X(280, 8, 8, 7, 5) X(300, 8, 10, 10, 3)                 \
  X(330, 8, 11, 10, 3) X(350, 8, 10, 7, 5) X(360, 8, 9, 8, 5) X(390, 8, 13, 10, 3) X(400, 8, 10, 10, 4)                \
  X(420, 8, 10, 7, 6) X(440, 8, 22, 20, 1) X(450, 8, 10, 9, 5) X(480, 8, 10, 8, 6) X(500, 8, 10, 10, 5)                \
  X(520, 8, 26, 20, 1) X(540, 8, 10, 9, 6) X(550, 8, 25, 22, 1) X(560, 8, 28, 20, 1) X(600, 8, 25, 24, 1)              \
  X(630, 8, 30, 21, 1) X(650, 8, 26, 25, 1) X(660, 8, 30, 22, 1) X(700, 8, 28, 25, 1) X(720, 8, 30, 24, 1)             \
  X(750, 8, 30, 25, 1) X(780, 8, 30, 26, 1) X(800, 8, 32, 25, 1) X(840, 8, 30, 28, 1) X(900, 8, 30, 30, 1)             \
  X(960, 8, 32, 30, 1)

// z passes (packed real rows of nz = 2 M points): X(M, rows per tile, R0, R1, R2); 256 threads
#define FG_SMOOTH_Z_PLANS(X) \
  /* p * 2^k lengths (run the tile kernels only where Fft3 prefers them to the sub-line kernels) */ \
  X(48, 64, 8, 6, 1) X(72, 32, 9, 8, 1) X(80, 32, 10, 8, 1) X(96, 32, 12, 8, 1) X(112, 32, 14, 8, 1) X(144, 16, 12, 12, 1) X(160, 16, 16, 10, 1) X(192, 16, 16, 12, 1) X(224, 16, 16, 14, 1) X(288, 8, 8, 6, 6) X(320, 8, 8, 8, 5) X(384, 8, 8, 8, 6) X(448, 8, 8, 8, 7) X(576, 4, 9, 8, 8) X(640, 4, 10, 8, 8) X(768, 4, 12, 8, 8) X(896, 4, 14, 8, 8) \
  X(25, 64, 5, 5, 1) X(30, 64, 6, 5, 1) X(35, 64, 7, 5, 1) X(45, 64, 9, 5, 1) X(50, 64, 10, 5, 1) X(60, 64, 10, 6, 1)  \
  X(65, 32, 13, 5, 1) X(70, 32, 10, 7, 1) X(75, 32, 15, 5, 1) X(90, 32, 10, 9, 1) X(100, 32, 10, 10, 1)                \
  X(105, 32, 15, 7, 1) X(110, 32, 22, 5, 1) X(120, 32, 15, 8, 1) X(125, 32, 5, 5, 5) X(130, 16, 13, 10, 1)             \
  X(135, 16, 15, 9, 1) X(140, 16, 14, 10, 1) X(150, 16, 15, 10, 1) X(165, 16, 15, 11, 1) X(175, 16, 7, 5, 5)           \
  X(180, 16, 15, 12, 1) X(195, 16, 15, 13, 1) X(200, 16, 8, 5, 5) X(210, 16, 15, 14, 1) X(220, 16, 22, 10, 1)          \
  X(225, 16, 15, 15, 1) X(240, 16, 16, 15, 1) X(250, 16, 10, 5, 5) X(260, 8, 13, 10, 2) X(270, 8, 9, 6, 5)             \
  X(275, 8, 11, 5, 5) X(280, 8, 8, 7, 5) X(300, 8, 10, 10, 3) X(315, 8, 9, 7, 5) X(325, 8, 13, 5, 5)                   \
  X(330, 8, 11, 10, 3) X(350, 8, 10, 7, 5) X(360, 8, 9, 8, 5) X(375, 8, 15, 5, 5) X(390, 8, 13, 10, 3)                 \
  X(400, 8, 10, 10, 4) X(420, 8, 10, 7, 6) X(440, 8, 22, 20, 1) X(450, 8, 10, 9, 5) X(480, 8, 10, 8, 6)                \
  X(500, 8, 10, 10, 5)

// fused x pass on the joint image of three components: X(N, columns per tile, threads, values per thread, R0, R1, R2)
#define FG_SMOOTH_X_PLANS(X) \
  /* p * 2^k lengths (run the tile kernels only where Fft3 prefers them to the sub-line kernels) */ \
  X(48, 16, 256, 20, 8, 6, 1) X(72, 16, 256, 20, 9, 8, 1) X(80, 16, 256, 20, 10, 8, 1) X(96, 16, 512, 20, 12, 8, 1) X(112, 8, 256, 20, 14, 8, 1) X(144, 8, 256, 20, 16, 9, 1) X(160, 8, 256, 20, 16, 10, 1) X(192, 8, 512, 20, 16, 12, 1) X(224, 8, 512, 20, 16, 14, 1) X(288, 8, 512, 20, 18, 16, 1) X(320, 8, 512, 20, 20, 16, 1) X(448, 4, 512, 20, 8, 8, 7) X(576, 4, 512, 20, 9, 8, 8) X(640, 4, 512, 20, 10, 8, 8) \
  X(50, 16, 256, 20, 10, 5, 1) X(60, 16, 256, 20, 10, 6, 1) X(70, 16, 256, 20, 10, 7, 1) X(90, 16, 256, 20, 10, 9, 1)  \
  X(100, 16, 256, 20, 10, 10, 1) X(110, 8, 256, 20, 11, 10, 1) X(120, 8, 256, 20, 12, 10, 1)                           \
  X(130, 8, 256, 20, 13, 10, 1) X(140, 8, 256, 20, 14, 10, 1) X(150, 8, 256, 20, 15, 10, 1)                            \
  X(180, 8, 256, 20, 18, 10, 1) X(200, 8, 256, 20, 20, 10, 1) X(210, 8, 512, 20, 15, 14, 1)                            \
  X(220, 8, 512, 20, 20, 11, 1) X(240, 8, 512, 20, 16, 15, 1) X(250, 8, 256, 32, 25, 10, 1)                            \
  X(260, 8, 512, 20, 20, 13, 1) X(270, 8, 512, 20, 18, 15, 1) X(280, 8, 512, 20, 20, 14, 1)                            \
  X(300, 8, 512, 20, 20, 15, 1) X(360, 8, 512, 20, 20, 18, 1) X(400, 8, 512, 20, 20, 20, 1)                            \
  X(420, 4, 512, 20, 10, 7, 6) X(440, 4, 512, 20, 11, 10, 4) X(450, 4, 512, 20, 10, 9, 5)                              \
  X(480, 4, 512, 20, 10, 8, 6) X(490, 4, 512, 20, 10, 7, 7) X(500, 4, 512, 20, 10, 10, 5)                              \
  X(520, 4, 512, 20, 13, 10, 4) X(540, 4, 512, 20, 10, 9, 6) X(560, 4, 512, 20, 10, 8, 7)                              \
  X(600, 4, 512, 20, 10, 10, 6) X(720, 4, 512, 20, 18, 10, 4) X(800, 4, 512, 20, 20, 10, 4)

// the one-component instance (heat / porous modes): the same fields
#define FG_SMOOTH_X1_PLANS(X) \
  X(50, 16, 256, 20, 10, 5, 1) X(60, 16, 256, 20, 10, 6, 1) X(70, 16, 256, 20, 10, 7, 1) X(90, 16, 256, 20, 10, 9, 1)  \
  X(100, 16, 256, 20, 10, 10, 1) X(110, 16, 256, 20, 11, 10, 1) X(120, 16, 256, 20, 12, 10, 1)                         \
  X(130, 16, 256, 20, 13, 10, 1) X(140, 16, 256, 20, 14, 10, 1) X(150, 16, 256, 20, 15, 10, 1)                         \
  X(180, 16, 256, 20, 15, 12, 1) X(200, 16, 256, 20, 20, 10, 1) X(210, 16, 256, 20, 15, 14, 1)                         \
  X(220, 16, 512, 20, 20, 11, 1) X(240, 16, 256, 20, 16, 15, 1) X(250, 16, 256, 32, 25, 10, 1)                         \
  X(260, 16, 512, 20, 20, 13, 1) X(270, 16, 512, 20, 18, 15, 1) X(280, 16, 512, 20, 20, 14, 1)                         \
  X(300, 16, 512, 20, 20, 15, 1) X(330, 8, 256, 32, 22, 15, 1) X(350, 8, 256, 32, 25, 14, 1)                           \
  X(360, 8, 256, 20, 20, 18, 1) X(390, 8, 256, 32, 26, 15, 1) X(400, 8, 256, 20, 20, 20, 1)                            \
  X(420, 8, 256, 32, 21, 20, 1) X(440, 8, 256, 32, 22, 20, 1) X(450, 8, 256, 32, 25, 18, 1)                            \
  X(480, 8, 256, 32, 24, 20, 1) X(500, 8, 256, 32, 25, 20, 1) X(520, 8, 256, 32, 26, 20, 1)                            \
  X(540, 8, 256, 32, 27, 20, 1) X(560, 8, 256, 32, 28, 20, 1) X(600, 8, 256, 32, 25, 24, 1)                            \
  X(720, 8, 256, 32, 30, 24, 1) X(800, 8, 256, 32, 32, 25, 1) X(900, 8, 256, 32, 30, 30, 1)                            \
  X(1000, 8, 512, 20, 10, 10, 10)

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
