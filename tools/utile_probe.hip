// Where does a marching step of the tiled displacement sweep (k_u_tile) spend its time?
//
// Compiles fibergen_amd/csrc/fg_kernels_fast.hip into this translation unit with -DFG_PROBE_K1 and prints the mean
// cycle count (s_memtime) between the marks of one step (step FG_PROBE_K1_STEP of the march) for one thread of
// every FG_PROBE_K1_STRIDE-th workgroup, plus the length of the whole march.  Development tool.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DFG_PROBE_K1 -I fibergen_amd/csrc \
//         tools/utile_probe.hip -o build_tools/utile_probe && build_tools/utile_probe 512
#ifndef FG_PROBE_K1_THREAD
#define FG_PROBE_K1_THREAD 64
#endif
#ifndef FG_PROBE_K1_STEP
#define FG_PROBE_K1_STEP 16
#endif
#ifndef FG_PROBE_K1_STRIDE
#define FG_PROBE_K1_STRIDE 96
#endif
#include "fg_kernels_fast.hip"

#include <cstdio>

using namespace fg;

__global__ void k_fill(double* x, long n, double lo, double hi) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = lo + (hi - lo) * (double)((i * 2654435761u) & 0xffff) / 65536.0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int rows = argc > 2 ? atoi(argv[2]) : 8;
  Grid g = make_grid(n, n, n, 1.0, 1.0, 1.0);
  hipStream_t s;
  FG_HIP_CHECK(hipStreamCreate(&s));
  double *u = nullptr, *f = nullptr, *mod = nullptr, *partial = nullptr, *sums = nullptr;
  FG_HIP_CHECK(hipMalloc(&u, 3 * g.n * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&f, 3 * g.n * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&mod, 2 * g.n * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&partial, 6 * (1 << 20) * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&sums, 16 * sizeof(double)));
  k_fill<<<(unsigned)((3 * g.n + 255) / 256), 256, 0, s>>>(u, 3 * g.n, -0.01, 0.01);
  k_fill<<<(unsigned)((2 * g.n + 255) / 256), 256, 0, s>>>(mod, 2 * g.n, 1.0, 10.0);
  FieldPtrs<3> up, fp;
  FieldPtrs<2> mp;
  for (int c = 0; c < 3; ++c) up.p[c] = u + c * g.n, fp.p[c] = f + c * g.n;
  mp.p[0] = mod;
  mp.p[1] = mod + g.n;
  Vec6 E;
  for (int c = 0; c < 6; ++c) E.v[c] = 0.1 * (c + 1);
  hipEvent_t e0, e1;
  FG_HIP_CHECK(hipEventCreate(&e0));
  FG_HIP_CHECK(hipEventCreate(&e1));
  const int reps = 5;
  for (int r = 0; r < 2; ++r) launch_u_tile(g, 2.0, 1.0, up, mp, fp, E, partial, sums, rows, s);
  FG_HIP_CHECK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) launch_u_tile(g, 2.0, 1.0, up, mp, fp, E, partial, sums, rows, s);
  FG_HIP_CHECK(hipEventRecord(e1, s));
  FG_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0;
  FG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%d^3 tiled sweep (rows %d): %.3f ms per launch (incl. the fold of the norms)\n", n, rows, ms / reps);
  static unsigned long long h[kK1ProbeBlocks][kK1ProbeSlots];
  FG_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_k1_probe), sizeof(h)));
  const char* what[9] = {"", "step start -> u in LDS", "barrier 1", "strain + polarisation", "tau in LDS",
                         "barrier 2", "divergence, stores, advance", "", ""};
  double d[9] = {0};
  double march = 0;
  int cnt = 0;
  for (int b = 0; b < kK1ProbeBlocks; ++b) {
    if (!h[b][0] || !h[b][8] || !h[b][7]) continue;
    ++cnt;
    for (int k = 1; k < 7; ++k) d[k] += (double)(h[b][k + 1] - h[b][k]);
    march += (double)(h[b][8] - h[b][0]);
  }
  printf("sampled workgroups: %d, thread %d, step %d\n", cnt, FG_PROBE_K1_THREAD, FG_PROBE_K1_STEP);
  double step = 0;
  for (int k = 1; k < 7; ++k) {
    printf("  %-30s %8.0f cycles\n", what[k], d[k] / cnt);
    step += d[k] / cnt;
  }
  printf("  one step %.0f cycles; whole march %.0f cycles\n", step, march / cnt);
  return 0;
}
