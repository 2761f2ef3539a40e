// Bytes-only probe: z and y transforms of one z-y plane fused in LDS (128^3: a complex plane is 128 x 72 x 16 B = 147 KB).
// One workgroup per (x plane, component): loads the real plane [ny][nzp] (nzp = 2 nzc doubles), stores the complex plane.
// Compare with the two separate passes (r2c_z 18.8 us + c2c_y 23.5 us at 128^3 in bench.py's kernel table).
//   hipcc -O3 --offload-arch=gfx950 tools/plane_probe.hip -o tools/build/plane_probe && tools/build/plane_probe 128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%d %s\n", __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_plane(const v2d* in, v2d* out, int plane16, int work) {
  extern __shared__ v2d lds[];
  const long base = (long)blockIdx.x * plane16 + (long)blockIdx.y * gridDim.x * plane16;
  for (int i = threadIdx.x; i < plane16; i += THREADS) lds[i] = in[base + i];
  __syncthreads();
  // stand-in for the butterflies: `work` dependent LDS round trips per element
  for (int w = 0; w < work; ++w) {
    for (int i = threadIdx.x; i < plane16; i += THREADS) {
      v2d a = lds[i], b = lds[(i + 64 * 9) % plane16];
      lds[i] = a * 0.5 + b * 0.5;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < plane16; i += THREADS) out[base + i] = lds[i];
}

__global__ void k_copy(const v2d* in, v2d* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128;
  const int nzc = ((n / 2 + 1 + 7) / 8) * 8;
  const int plane16 = n * nzc;             // 16-byte elements per plane
  const long total = (long)n * plane16 * 3;
  v2d *a, *b;
  CK(hipMalloc(&a, total * 16));
  CK(hipMalloc(&b, total * 16));
  CK(hipMemset(a, 1, total * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t lds = (size_t)plane16 * 16;
  printf("n = %d: plane %zu KB, %d workgroups, %.1f MB per pass\n", n, lds / 1024, 3 * n, 2.0 * total * 16 / 1e6);
  auto time_it = [&](auto launch, const char* name) {
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / 20 < best) best = ms / 20;
    }
    CK(hipGetLastError());
    printf("%-48s %.2f us  %.0f GB/s\n", name, best * 1e3, 2.0 * total * 16 / best / 1e6);
  };
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_plane<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_plane<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  time_it([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, total); }, "plain copy (2048 x 256)");
  for (int work : {0, 2, 6}) {
    char nm[96];
    snprintf(nm, sizeof nm, "plane in LDS, 512 threads, %d LDS sweeps", work);
    time_it([&] { hipLaunchKernelGGL(k_plane<512>, dim3(n, 3), dim3(512), lds, 0, a, b, plane16, work); }, nm);
    snprintf(nm, sizeof nm, "plane in LDS, 1024 threads, %d LDS sweeps", work);
    time_it([&] { hipLaunchKernelGGL(k_plane<1024>, dim3(n, 3), dim3(1024), lds, 0, a, b, plane16, work); }, nm);
  }
  return 0;
}
