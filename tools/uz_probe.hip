// Where does a marching step of the z-attached displacement sweep (k_uz_tile) spend its time?
//
// Compiles fibergen_amd/csrc/fg_kernels_zsweep.hip into this translation unit with -DFG_PROBE_UZ and prints, for three waves
// (wave 0: c2r role only; a middle wave: both transform roles; the last wave: r2c role only) of every FG_PROBE_UZ_STRIDE-th
// workgroup, the mean cycle counts between the marks of step FG_PROBE_UZ_STEP of the march, and the kernel's time on a
// synthetic field.  Development tool.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DFG_PROBE_UZ -I fibergen_amd/csrc \
//         tools/uz_probe.hip -o build_tools/uz_probe && build_tools/uz_probe 256
#ifndef FG_PROBE_UZ_STEP
#define FG_PROBE_UZ_STEP 16
#endif
#ifndef FG_PROBE_UZ_STRIDE
#define FG_PROBE_UZ_STRIDE 4
#endif
#include "fg_kernels_zsweep.hip"

#include <cstdio>
#include <vector>

#include "fg_fft_tables.h"

using namespace fg;

__global__ void k_fill(double* x, long n, double lo, double hi) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = lo + (hi - lo) * (double)((i * 2654435761u) & 0xffff) / 65536.0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  Grid g = make_grid(n, n, n, 1.0, 1.0, 1.0);
  hipStream_t s;
  FG_HIP_CHECK(hipStreamCreate(&s));
  double *u = nullptr, *f = nullptr, *mod = nullptr, *partial = nullptr, *sums = nullptr;
  FG_HIP_CHECK(hipMalloc(&u, 3 * g.n * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&f, 3 * g.n * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&mod, 2 * g.n * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&partial, 12 * (1 << 16) * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&sums, 16 * sizeof(double)));
  k_fill<<<(unsigned)((3 * g.n + 255) / 256), 256, 0, s>>>(u, 3 * g.n, -0.01, 0.01);
  k_fill<<<(unsigned)((2 * g.n + 255) / 256), 256, 0, s>>>(mod, 2 * g.n, 0.0, 1.0);
  std::vector<cplx> tw = make_pass_twiddles4(n / 2), wz = make_unit_roots(n, n / 2 + 1);
  cplx *dtw = nullptr, *dwz = nullptr;
  FG_HIP_CHECK(hipMalloc(&dtw, tw.size() * sizeof(cplx)));
  FG_HIP_CHECK(hipMalloc(&dwz, wz.size() * sizeof(cplx)));
  FG_HIP_CHECK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice));
  FG_HIP_CHECK(hipMemcpy(dwz, wz.data(), wz.size() * sizeof(cplx), hipMemcpyHostToDevice));
  FieldPtrs<3> up, fp;
  FieldPtrs<2> mp;
  for (int c = 0; c < 3; ++c) up.p[c] = u + c * g.n, fp.p[c] = f + c * g.n;
  mp.p[0] = mod;
  mp.p[1] = nullptr;
  PhaseTable pt = {};
  pt.n = 2;
  pt.mu[0] = 0.4; pt.mu[1] = 4.0; pt.lambda[0] = 0.6; pt.lambda[1] = 2.8;
  Vec6 E;
  for (int c = 0; c < 6; ++c) E.v[c] = 0.1 * (c + 1);
  hipEvent_t e0, e1;
  FG_HIP_CHECK(hipEventCreate(&e0));
  FG_HIP_CHECK(hipEventCreate(&e1));
  const int reps = 10;
  for (int r = 0; r < 2; ++r) launch_uz_tile(g, 2.0, 1.0, up, mp, fp, E, partial, sums, s, false, &pt, dtw, dwz);
  FG_HIP_CHECK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) launch_uz_tile(g, 2.0, 1.0, up, mp, fp, E, partial, sums, s, false, &pt, dtw, dwz);
  FG_HIP_CHECK(hipEventRecord(e1, s));
  FG_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0;
  FG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 56.0 * (double)n * n * n;
  printf("%d^3 z-attached sweep: %.4f ms per launch (incl. the fold of the norms), %.2f TB/s of its algorithmic 56 B/voxel\n", n,
         ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
  static unsigned long long h[kUzProbeBlocks][3][kUzProbeSlots];
  FG_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_uz_probe), sizeof(h)));
  const char* what[6] = {"B1: strain + polarisation", "barrier 1", "B2: divergence -> F image", "barrier 2",
                         "c2r of plane q+2 | r2c rounds", "barrier 3"};
  const char* wave[3] = {"wave 0 (c2r role)", "middle wave (c2r role)", "last wave (r2c role)"};
  for (int k = 0; k < 3; ++k) {
    double d[6] = {0};
    int cnt = 0;
    for (int b = 0; b < kUzProbeBlocks; ++b) {
      if (!h[b][k][0] || !h[b][k][6]) continue;
      ++cnt;
      for (int j = 0; j < 6; ++j) d[j] += (double)(h[b][k][j + 1] - h[b][k][j]);
    }
    if (!cnt) continue;
    double step = 0;
    printf("%s, %d workgroups sampled, step %d:\n", wave[k], cnt, FG_PROBE_UZ_STEP);
    for (int j = 0; j < 6; ++j) {
      printf("  %-32s %8.0f cycles\n", what[j], d[j] / cnt);
      step += d[j] / cnt;
    }
    printf("  one step %.0f cycles\n", step);
    if (k == 2) {
      double g = 0, ph = 0, stq = 0;
      int c2 = 0;
      for (int b = 0; b < kUzProbeBlocks; ++b) {
        if (!h[b][2][8] || !h[b][2][9]) continue;
        ++c2;
        g += (double)(h[b][2][8] - h[b][2][4]);
        ph += (double)(h[b][2][9] - h[b][2][8]);
        stq += (double)(h[b][2][5] - h[b][2][9]);
      }
      if (c2) printf("  r2c detail: gather %.0f, line phases %.0f, split + stores %.0f cycles\n", g / c2, ph / c2, stq / c2);
    }
  }
  return 0;
}
