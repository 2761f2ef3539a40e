// Linear byte mover with the stream count of the tiled sweep (4 inputs, 3 outputs, 16 B per lane): what the memory
// system gives a kernel with K1's bytes and no stencil at all.  Measured at 256^3 sizes: 5.4-5.6 TB/s (0.177-0.186 ms for
// 0.997 GB); the sweep itself: 0.978 GB in 0.188-0.193 ms.   hipcc -O3 --offload-arch=gfx950 tools/streams_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%d %s\n", __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
// 4 input streams, 3 output streams (the sweep's u0, u1, u2, phi -> f0, f1, f2), linear, 16 B per lane
__global__ __launch_bounds__(256) void k7(const v2d* a, const v2d* b, const v2d* c, const v2d* d, v2d* x, v2d* y, v2d* z, long n, int nt) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    v2d va = a[i], vb = b[i], vc = c[i], vd = d[i];
    if (nt) { __builtin_nontemporal_store(va + vd, &x[i]); __builtin_nontemporal_store(vb + vd, &y[i]); __builtin_nontemporal_store(vc + vd, &z[i]); }
    else { x[i] = va + vd; y[i] = vb + vd; z[i] = vc + vd; }
  }
}
int main() {
  const long n = 256L * 256 * 272 / 2;   // pairs of one padded component at 256^3
  v2d* p[7];
  for (int i = 0; i < 7; ++i) { CK(hipMalloc(&p[i], n * 16)); CK(hipMemset(p[i], 0, n * 16)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nt = 0; nt < 2; ++nt) for (int blocks : {2048, 8192, 65536}) {
    CK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k7, dim3(blocks), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], p[6], n, nt);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("7 streams (4 in, 3 out), %s stores, %d blocks: %.3f ms, %.0f GB/s\n", nt ? "streaming" : "plain", blocks, ms / 20, 7.0 * n * 16 / (ms / 20) / 1e6);
  }
}
