// Does a strided-pass tile (8 columns x N rows of 16-byte complex values, row stride `ls`) load / store as fast when a
// WAVE owns whole lines (lanes = rows of one or two columns: 16- or 32-byte pieces of 32 to 64 different 128-byte lines
// per wave instruction, the other waves of the workgroup touching the rest of the same lines) as with the present mapping
// (8 lanes = the 8 columns of one row: whole 128-byte segments)?  With wave-owned lines the exchanges of the y and fused x
// passes would need no workgroup barrier.  Bytes only: every thread loads its 8 points, then stores them (streaming).
//   hipcc -O3 --offload-arch=gfx950 tools/tile_access_probe.hip -o /tmp/tile_access_probe && /tmp/tile_access_probe [n]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%d %s\n", __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// N rows, T = N / 8 threads per column, 8 columns per tile; MODE 0: tid = jt * 8 + t (rows x columns: coalesced segments),
// MODE 1: tid = t * T + jt (a column's T threads are consecutive lanes)
template <int N, int MODE, int C = 8>
__global__ __launch_bounds__(N / 8 * C) void k_tile(const v2d* in, v2d* out, long ls, int tiles_per_outer, long os) {
  constexpr int T = N / 8;
  const int tid = threadIdx.x;
  const int t = MODE == 0 ? tid % C : tid / T;
  const int jt = MODE == 0 ? tid / C : tid % T;
  const long base = (long)(blockIdx.x / tiles_per_outer) * os + (long)(blockIdx.x % tiles_per_outer) * C + t;
  v2d v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(&in[base + (long)(jt + q * T) * ls]);
#pragma unroll
  for (int q = 0; q < 8; ++q) __builtin_nontemporal_store(v[q], &out[base + (long)(jt + q * T) * ls]);
}

template <int N>
void run(int n) {
  // the y pass of an n^3 grid: lines along y (stride nzc), columns = kz, outer = x
  const int nzc = ((n / 2 + 1 + 7) / 8) * 8;
  const long total = (long)n * n * nzc;
  v2d *a, *b;
  CK(hipMalloc(&a, total * 16)); CK(hipMalloc(&b, total * 16));
  CK(hipMemset(a, 1, total * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tpo = nzc / 8;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 10; ++it) {
        if (mode == 0) hipLaunchKernelGGL((k_tile<N, 0>), dim3(tpo * n), dim3(N), 0, 0, a, b, (long)nzc, tpo, (long)n * nzc);
        else hipLaunchKernelGGL((k_tile<N, 1>), dim3(tpo * n), dim3(N), 0, 0, a, b, (long)nzc, tpo, (long)n * nzc);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("n %d y-pass tiles, %s: %.3f ms, %.0f GB/s\n", n, mode == 0 ? "lanes = columns (128-byte segments)" : "a wave owns lines (32-byte pieces)  ",
             ms / 10, 2.0 * total * 16 / (ms / 10) / 1e6);
    }
  }
  // the x pass: lines along x (stride ny * nzc), columns = (y, kz) flattened
  const long ls = (long)n * nzc;
  const int tpo2 = (int)(ls / 8);
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) {
      if (mode == 0) hipLaunchKernelGGL((k_tile<N, 0>), dim3(tpo2), dim3(N), 0, 0, a, b, ls, tpo2, 0L);
      else hipLaunchKernelGGL((k_tile<N, 1>), dim3(tpo2), dim3(N), 0, 0, a, b, ls, tpo2, 0L);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("n %d x-pass tiles, %s: %.3f ms, %.0f GB/s\n", n, mode == 0 ? "lanes = columns (128-byte segments)" : "a wave owns lines                   ",
           ms / 10, 2.0 * total * 16 / (ms / 10) / 1e6);
  }
  // the x pass with 16-column (256-byte) and 4-column (64-byte) segments
  for (int c = 0; c < 2; ++c) {
    const int C = c == 0 ? 16 : 4;
    const int tp = (int)(ls / C);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) {
      if (c == 0) hipLaunchKernelGGL((k_tile<N, 0, 16>), dim3(tp), dim3(N / 8 * 16), 0, 0, a, b, ls, tp, 0L);
      else hipLaunchKernelGGL((k_tile<N, 0, 4>), dim3(tp), dim3(N / 8 * 4), 0, 0, a, b, ls, tp, 0L);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("n %d x-pass tiles of %d columns (%d-byte segments): %.3f ms, %.0f GB/s\n", n, C, C * 16, ms / 10, 2.0 * total * 16 / (ms / 10) / 1e6);
  }
  CK(hipFree(a)); CK(hipFree(b));
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  if (n == 256) run<256>(n);
  else if (n == 512) run<512>(n);
  else if (n == 128) run<128>(n);
  return 0;
}
