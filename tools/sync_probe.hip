// Flag hand-over between workgroups of one XCD: workgroup b waits for the flag of workgroup b - 8
// (round-robin dispatch puts them on the same XCD), with relaxed / acquire device-scope loads or a read-modify-write as
// the poll and a read-modify-write or a store as the signal.  Measured: 0.8 us per hop once warm, every combination works.
//   hipcc -O3 --offload-arch=gfx950 tools/sync_probe.hip -o /tmp/sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%d %s\n", __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
// WG b (b >= 8) waits until flag[b - 8] != 0 (same XCD under round-robin dispatch), then sets flag[b]
template <int MODE>
__global__ void k(int* flag, long long* spins_out) {
  const int b = blockIdx.x;
  long long spins = 0;
  if (threadIdx.x == 0) {
    if (b >= 8) {
      int* p = &flag[b - 8];
      for (;;) {
        int v;
        if (MODE == 0 || MODE == 3 || MODE == 4) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) v = __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        else v = __hip_atomic_fetch_add(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) break;
        if (++spins > 2000000) { spins = -1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (MODE == 3) __hip_atomic_store(&flag[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 4) __hip_atomic_store(&flag[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(&flag[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spins_out[b] = spins;
  }
}
template <int MODE>
void run(const char* name, int n) {
  int* flag; long long* sp;
  CK(hipMalloc(&flag, n * 4)); CK(hipMalloc(&sp, n * 8));
  CK(hipMemset(flag, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(64), 0, 0, flag, sp);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long* h = (long long*)malloc(n * 8); CK(hipMemcpy(h, sp, n * 8, hipMemcpyDeviceToHost));
  long long mx = 0, bad = 0; for (int i = 0; i < n; ++i) { if (h[i] < 0) bad++; if (h[i] > mx) mx = h[i]; }
  printf("%s: %d workgroups, chain depth %d: %.3f ms (%.2f us per hop), max spins %lld, gave up %lld\n", name, n, n / 8, ms, 1e3 * ms / (n / 8), mx, bad);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<0>("relaxed load sc1        ", 512);
  run<1>("acquire load            ", 512);
  run<2>("atomic add 0 (rmw)      ", 512);
  run<3>("relaxed STORE + relaxed load", 512);
  run<4>("release STORE + relaxed load", 512);
  return 0;
}
