// Can the 256 MB Infinity Cache carry the hand-over between two x-plane-local passes if they run chunk by chunk?
// (VERDICT r5 "next" #1.)  Bytes only; a pass = read 32 KB, write 32 KB per workgroup, as tools/mall_probe.hip.
//  (a) capacity knee: an in-place read-modify-write chain over W MB, W = 32 ... 384
//  (b) a field of S MB processed as A(all) -> B(all) against A(chunk c) -> B(chunk c), c = 0 .. S/chunk, both in place
//      (the z pass of the transform chain is in place) and out of place for B (the y pass writing a second buffer);
//      chunked once as 2 launches per chunk and once as ONE launch whose workgroups are ordered
//      A(c0) B(c0) A(c1) B(c1) ... with B(c) spinning on a completion counter of A(c) (in-order dispatch).
//   hipcc -O3 --offload-arch=gfx950 tools/mall_chunk_probe.hip -o tools/build/mall_chunk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%d %s\n", __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int NTL, int NTS>
__device__ __forceinline__ void move32k(const v2d* __restrict__ s, v2d* __restrict__ d) {
  v2d v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = NTL ? __builtin_nontemporal_load(&s[q * 256]) : s[q * 256];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (NTS) __builtin_nontemporal_store(v[q] + v[q], &d[q * 256]);
    else d[q * 256] = v[q] + v[q];
  }
}

// blocks [first, first + gridDim.x) of the field
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_pass(const v2d* a, v2d* b, long first) {
  const long c = first + blockIdx.x;
  move32k<NTL, NTS>(a + c * 2048 + threadIdx.x, b + c * 2048 + threadIdx.x);
}

// ONE launch: groups of 2*cb workgroups; the first cb run pass A on chunk g (src -> mid), the second cb pass B (mid -> dst)
// after all of A(g) has finished.  B's block j reads what A's block (j + cb/2) % cb wrote: a cross-workgroup dependency as
// between a z pass (rows) and a y pass (columns) of one x plane.
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_pair(const v2d* src, v2d* mid, v2d* dst, long nblocks, int cb, unsigned* done, unsigned epoch) {
  const long g = blockIdx.x / (2 * cb);
  const int r = blockIdx.x % (2 * cb);
  const long base = g * cb;
  const int here = (int)((nblocks - base) < cb ? (nblocks - base) : cb);   // blocks of this chunk
  if (r < cb) {
    if (r >= here) return;
    const long c = base + r;
    move32k<NTL, 0>(src + c * 2048 + threadIdx.x, mid + c * 2048 + threadIdx.x);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&done[g], 1u);
  } else {
    const int j = r - cb;
    if (j >= here) return;
    if (threadIdx.x == 0) {
      const unsigned want = epoch * (unsigned)here;
      while (__hip_atomic_load(&done[g], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const long c = base + (j + here / 2) % here;
    move32k<0, NTS>(mid + c * 2048 + threadIdx.x, dst + c * 2048 + threadIdx.x);
  }
}

static hipEvent_t e0, e1;
template <class F> static float timed(int warm, int reps, F f) {
  for (int i = 0; i < warm; ++i) f(i);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f(warm + i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long maxmb = 3245;
  const long maxblocks = maxmb * 1024 * 1024 / 32768;
  v2d *a, *b;
  CK(hipMalloc(&a, maxblocks * 32768)); CK(hipMalloc(&b, maxblocks * 32768));
  CK(hipMemset(a, 0, maxblocks * 32768)); CK(hipMemset(b, 0, maxblocks * 32768));
  unsigned* done; CK(hipMalloc(&done, 1 << 20)); CK(hipMemset(done, 0, 1 << 20));

  printf("(a) capacity knee: in-place read-modify-write chain, plain loads / stores\n");
  for (long w = 32; w <= 384; w += 32) {
    const long nb = w * 1024 * 1024 / 32768;
    float ms = timed(4, 40, [&](int) { hipLaunchKernelGGL((k_pass<0, 0>), dim3(nb), dim3(256), 0, 0, a, a, 0L); });
    printf("  W = %3ld MB: %.4f ms per pass, %.0f GB/s\n", w, ms, 2.0 * nb * 32768 / ms / 1e6);
  }
  printf("(a') two buffers a -> b -> a, W = both\n");
  for (long w = 64; w <= 384; w += 64) {
    const long nb = w / 2 * 1024 * 1024 / 32768;
    float ms = timed(4, 40, [&](int i) { hipLaunchKernelGGL((k_pass<0, 0>), dim3(nb), dim3(256), 0, 0, i & 1 ? b : a, i & 1 ? a : b, 0L); });
    printf("  W = %3ld MB: %.4f ms per pass, %.0f GB/s\n", w, ms, 2.0 * nb * 32768 / ms / 1e6);
  }

  const long sizes[2] = {417, 3245};
  const int chunks[] = {0, 8, 16, 32, 48, 64, 96, 128};
  for (int rep = 0; rep < 2; ++rep)
    for (long S : sizes) {
      const long nb = S * 1024 * 1024 / 32768;
      const int reps = S > 1000 ? 6 : 20;
      printf("(b) field of %ld MB, pass A then pass B (ms for the PAIR; 4 x S bytes):\n", S);
      for (int oop = 0; oop < 2; ++oop)       // 0: a -> a -> a;  1: a -> a -> b (B writes the other buffer)
        for (int nt = 0; nt < 2; ++nt)        // 1: streaming on the outside of the pair (A's loads, B's stores)
          for (int ck : chunks) {
            v2d* dst = oop ? b : a;
            if (ck == 0) {
              float ms = timed(2, reps, [&](int) {
                if (nt) { hipLaunchKernelGGL((k_pass<1, 0>), dim3(nb), dim3(256), 0, 0, a, a, 0L); hipLaunchKernelGGL((k_pass<0, 1>), dim3(nb), dim3(256), 0, 0, a, dst, 0L); }
                else    { hipLaunchKernelGGL((k_pass<0, 0>), dim3(nb), dim3(256), 0, 0, a, a, 0L); hipLaunchKernelGGL((k_pass<0, 0>), dim3(nb), dim3(256), 0, 0, a, dst, 0L); }
              });
              printf("  %s %s whole field        : %.4f ms, %.0f GB/s\n", oop ? "a->a->b" : "in place", nt ? "nt-outside" : "plain     ", ms, 4.0 * nb * 32768 / ms / 1e6);
              continue;
            }
            const int cb = ck * 1024 * 1024 / 32768;
            const long ng = (nb + cb - 1) / cb;
            float ms = timed(2, reps, [&](int) {
              for (long g = 0; g < ng; ++g) {
                const long first = g * cb; const long n = (nb - first) < cb ? (nb - first) : cb;
                if (nt) { hipLaunchKernelGGL((k_pass<1, 0>), dim3(n), dim3(256), 0, 0, a, a, first); hipLaunchKernelGGL((k_pass<0, 1>), dim3(n), dim3(256), 0, 0, a, dst, first); }
                else    { hipLaunchKernelGGL((k_pass<0, 0>), dim3(n), dim3(256), 0, 0, a, a, first); hipLaunchKernelGGL((k_pass<0, 0>), dim3(n), dim3(256), 0, 0, a, dst, first); }
              }
            });
            unsigned epoch = 0;
            CK(hipMemset(done, 0, 1 << 20));
            float mf = timed(2, reps, [&](int) {
              ++epoch;
              if (nt) hipLaunchKernelGGL((k_pair<1, 1>), dim3(ng * 2 * cb), dim3(256), 0, 0, a, a, dst, nb, cb, done, epoch);
              else    hipLaunchKernelGGL((k_pair<0, 0>), dim3(ng * 2 * cb), dim3(256), 0, 0, a, a, dst, nb, cb, done, epoch);
            });
            printf("  %s %s chunks of %3d MB (%3ld): launches %.4f ms, %.0f GB/s | one launch %.4f ms, %.0f GB/s\n", oop ? "a->a->b" : "in place",
                   nt ? "nt-outside" : "plain     ", ck, ng, ms, 4.0 * nb * 32768 / ms / 1e6, mf, 4.0 * nb * 32768 / mf / 1e6);
          }
    }
  return 0;
}
