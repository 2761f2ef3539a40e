// Does the 256 MB Infinity Cache hand data from one pass to the next when the second pass starts where the first one ended?
// A chain of in-place style passes over a field of `mb` MB (each: read a -> write b, block i handles chunk i of the field in
// launch order); every second pass walks the field either in the same direction as its predecessor or in the opposite one.
// Bytes only.   hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o tools/build/mall_probe;  mall_probe [mb=417]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%d %s\n", __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
// chunk = 256 threads x 8 x 16 B = 32 KB per block
__global__ __launch_bounds__(256) void k_pass(const v2d* a, v2d* b, long nchunks, int reverse, int nt) {
  const long c = reverse ? nchunks - 1 - blockIdx.x : blockIdx.x;
  const v2d* s = a + c * 2048 + threadIdx.x;
  v2d* d = b + c * 2048 + threadIdx.x;
  v2d v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = nt ? __builtin_nontemporal_load(&s[q * 256]) : s[q * 256];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (nt) __builtin_nontemporal_store(v[q] + v[q], &d[q * 256]);
    else d[q * 256] = v[q] + v[q];
  }
}
int main(int argc, char** argv) {
  const long mb = argc > 1 ? atol(argv[1]) : 417;
  const long nchunks = mb * 1024 * 1024 / 32768;
  v2d *a, *b;
  CK(hipMalloc(&a, nchunks * 32768)); CK(hipMalloc(&b, nchunks * 32768));
  CK(hipMemset(a, 0, nchunks * 32768)); CK(hipMemset(b, 0, nchunks * 32768));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (int nt = 0; nt < 2; ++nt)
      for (int alt = 0; alt < 2; ++alt) {
        for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(k_pass, dim3(nchunks), dim3(256), 0, 0, w & 1 ? b : a, w & 1 ? a : b, nchunks, alt ? (w & 1) : 0, nt);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 40; ++it) hipLaunchKernelGGL(k_pass, dim3(nchunks), dim3(256), 0, 0, it & 1 ? b : a, it & 1 ? a : b, nchunks, alt ? (it & 1) : 0, nt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%ld MB field, %s loads/stores, %s: %.4f ms per pass, %.0f GB/s\n", mb, nt ? "streaming" : "plain",
               alt ? "alternating directions" : "same direction      ", ms / 40, 2.0 * nchunks * 32768 / (ms / 40) / 1e6);
      }
}
