// Where does a tile of the fused x pass (x-FFT -> Green operator -> x-iFFT) spend its time?
//
// Compiles fibergen_amd/csrc/fg_fft.hip into this one translation unit with -DFG_PROBE, runs the pass on a
// random 3-component half spectrum and prints, per phase boundary, the mean cycle count (s_memtime) since the
// tile started, for one thread of every 256th workgroup.  Development tool, not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DFG_PROBE -DFG_PROBE_THREAD=0 \
//         -I fibergen_amd/csrc tools/xfused_probe.hip -o gpurun_out/xfused_probe && gpurun_out/xfused_probe 512
#ifndef FG_PROBE_THREAD
#define FG_PROBE_THREAD 0
#endif
#include "fg_fft.hip"

#include <cstdio>
#include <vector>

using namespace fg;

__global__ void k_fill(double* x, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int ny = argc > 2 ? atoi(argv[2]) : n, nz = argc > 3 ? atoi(argv[3]) : n;
  const int reps = 5;
  Grid g = make_grid(n, ny, nz, 1.0, 1.0, 1.0);
  hipStream_t s;
  FG_HIP_CHECK(hipStreamCreate(&s));
  Fft3 fft(g, s);
  double* data = nullptr;
  FG_HIP_CHECK(hipMalloc(&data, 3 * g.n * sizeof(double)));
  k_fill<<<(unsigned)((3 * g.n + 255) / 256), 256, 0, s>>>(data, 3 * g.n);
  G0Params gp;
  const int len[3] = {g.nx, g.ny, g.nzc};
  for (int a = 0; a < 3; ++a) {
    std::vector<double> kpm(len[a], 1.0);
    std::vector<cplx> kp(len[a], cmake(0.6, 0.8));
    double* dk = nullptr;
    cplx* dc = nullptr;
    FG_HIP_CHECK(hipMalloc(&dk, len[a] * sizeof(double)));
    FG_HIP_CHECK(hipMalloc(&dc, len[a] * sizeof(cplx)));
    FG_HIP_CHECK(hipMemcpy(dk, kpm.data(), len[a] * sizeof(double), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(dc, kp.data(), len[a] * sizeof(cplx), hipMemcpyHostToDevice));
    gp.kpm[a] = dk;
    gp.kp[a] = dc;
  }
  gp.c10 = -0.5;
  gp.c20 = 0.25;
  gp.inv_h0 = 2.0 * n;
  hipEvent_t e0, e1;
  FG_HIP_CHECK(hipEventCreate(&e0));
  FG_HIP_CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 2; ++r) fft.fused_g0(data, g.n, 0, 1.0 / n, gp, 0, 3);
  FG_HIP_CHECK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) fft.fused_g0(data, g.n, 0, 1.0 / n, gp, 0, 3);
  FG_HIP_CHECK(hipEventRecord(e1, s));
  FG_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0;
  FG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%d x %d x %d fused x pass: %.3f ms per launch, %.2f TB/s\n", n, ny, nz, ms / reps,
         2.0 * 3 * g.n * 8 / (ms / reps * 1e-3) / 1e12);

  static unsigned long long h[kProbeBlocks][kProbeSlots];
  FG_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe), sizeof(h)));
  double mean[kProbeSlots] = {0};
  int cnt = 0;
  int nslots = 0;
  for (int b = 0; b < kProbeBlocks; ++b) {
    if (!h[b][0]) continue;
    ++cnt;
    for (int p = 0; p < kProbeSlots; ++p) {
      if (!h[b][p]) break;
      mean[p] += (double)(h[b][p] - h[b][0]);
      if (p + 1 > nslots) nslots = p + 1;
    }
  }
  printf("sampled workgroups: %d, thread %d; cycles since tile start at the start of each phase (last = end)\n", cnt,
         FG_PROBE_THREAD);
  double prev = 0;
  for (int p = 0; p < nslots; ++p) {
    const double m = mean[p] / cnt;
    printf("  phase %2d  t=%9.0f  (+%7.0f)\n", p, m, m - prev);
    prev = m;
  }
  return 0;
}
