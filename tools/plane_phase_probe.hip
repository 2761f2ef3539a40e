// Where does a plane kernel (fg_fft_plane.h: z + y transforms of a z-y plane in one workgroup) spend its time?  Cycle stamps of
// one thread of every 16th workgroup at every phase boundary (the FG_PROBE marks of fg_fft.hip).  Development tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DFG_PROBE -DFG_PROBE_STRIDE=16 -DFG_PROBE_THREAD=0 \
//         -I fibergen_amd/csrc tools/plane_phase_probe.hip -o tools/build/plane_phase_probe && tools/build/plane_phase_probe 128 -1
#ifndef FG_PROBE_THREAD
#define FG_PROBE_THREAD 0
#endif
#include "fg_fft.hip"

#include <cstdio>

using namespace fg;

__global__ void k_fill(double* x, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128;
  const int dir = argc > 2 ? atoi(argv[2]) : -1;
  const int ncomp = argc > 3 ? atoi(argv[3]) : 3;   // workgroups = n * ncomp (round quantisation on 256 CUs)
  Grid g = make_grid(n, n, n, 1.0, 1.0, 1.0);
  hipStream_t s;
  FG_HIP_CHECK(hipStreamCreate(&s));
  Fft3 fft(g, s);
  double* data = nullptr;
  FG_HIP_CHECK(hipMalloc(&data, 4 * g.n * sizeof(double)));
  k_fill<<<(unsigned)((4 * g.n + 255) / 256), 256, 0, s>>>(data, 4 * g.n);
  hipEvent_t e0, e1;
  FG_HIP_CHECK(hipEventCreate(&e0));
  FG_HIP_CHECK(hipEventCreate(&e1));
  const int reps = 20;
  for (int r = 0; r < 3; ++r) fft.zy_plane(data, ncomp, g.n, dir);
  FG_HIP_CHECK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) fft.zy_plane(data, ncomp, g.n, dir);
  FG_HIP_CHECK(hipEventRecord(e1, s));
  FG_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0;
  FG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%d^3 plane kernel dir %d, %d components (%d workgroups): %.2f us per launch\n", n, dir, ncomp, n * ncomp, 1e3 * ms / reps);
  static unsigned long long h[kProbeBlocks][kProbeSlots];
  FG_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe), sizeof(h)));
  double mean[kProbeSlots] = {0};
  int cnt = 0, nslots = 0;
  for (int b = 0; b < kProbeBlocks; ++b) {
    if (!h[b][0]) continue;
    ++cnt;
    for (int p = 0; p < kProbeSlots; ++p) {
      if (!h[b][p]) break;
      mean[p] += (double)(h[b][p] - h[b][0]);
      if (p + 1 > nslots) nslots = p + 1;
    }
  }
  printf("sampled workgroups: %d; cycles since the start at the start of each phase (last = end)\n", cnt);
  double prev = 0;
  for (int p = 0; p < nslots; ++p) {
    const double m = mean[p] / cnt;
    printf("  phase %2d  t=%9.0f  (+%7.0f)\n", p, m, m - prev);
    prev = m;
  }
  return 0;
}
