// Feasibility probe for fusing the z and the y FFT pass through the L2 of an XCD.
//
// Question: if the workgroups of ONE XCD hand a z-y plane of one component from a row-wise stage (A: the z transform) to a
// column-wise stage (B: the y transform) through a small scratch buffer, does the hand-over stay in that XCD's 4 MB L2 --
// i.e. does the pair cost one read + one write of the field instead of two of each?
//
// The probe moves bytes only (no transforms): A copies 16 rows of a plane into the scratch, B copies 8-column tiles of the
// scratch to the output, with the access patterns of R2CKernel / StridedKernel.  Teams are formed from the hardware XCC id,
// work is handed out by per-XCD tickets (sequence s = one plane: nA row items, then nB column items; B(s) waits for all of
// A(s), A(s) for all of B(s-2): the scratch is double-buffered).  Release: stores + s_waitcnt vmcnt(0) + relaxed
// agent-scope atomic (no L2 write-back); acquire: spin on the atomic + buffer_inv sc1 (drops the L1).
// Baseline: the same two stages as two kernels through a full-size intermediate field.
//
//   hipcc -O3 --offload-arch=gfx950 tools/l2_probe.hip -o tools/l2_probe && tools/l2_probe [n]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));     \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

#ifndef TIMING
#define TIMING 0   // 1: in-kernel timers (their contended atomics cost more than the work: diagnosis only)
#endif
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ unsigned long long g_time2[8];   // A: load wait, store wait; B: load wait, store wait; counts A, B

struct Dims {
  int nx, ny, nzc;     // complex row pitch nzc
  int ncomp;
  int rows_per_item;   // A item: rows_per_item rows
  int nA, nB;          // items per plane
  long plane;          // complex per plane = ny * nzc
};

struct Sync {
  unsigned long long* ticket;     // [8]
  unsigned long long* next_unit;  // [1]
  int* unit_id;                   // [8][units]  (id + 1; 0 = not published)
  int* doneA;                     // [8][units]
  int* doneB;                     // [8][units]
  int units;
};

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ void item_A(const Dims& d, const v2d* in_plane, v2d* dst_plane, int item, int tid, bool nt_in) {
  // rows [item*R, item*R+R) of the plane: contiguous R*nzc complex
  const long base = (long)item * d.rows_per_item * d.nzc;
  const int n = d.rows_per_item * d.nzc;
  // all loads of the thread in flight before the first store (as in the transform kernels: 8 points per thread)
  const unsigned long long t0 = wall_clock64();
  v2d v[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int i = tid + q * 256;
    v[q] = i < n ? (nt_in ? __builtin_nontemporal_load(&in_plane[base + i]) : in_plane[base + i]) : (v2d){0, 0};
  }
  if (TIMING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long tl = wall_clock64();
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int i = tid + q * 256;
    if (i < n) dst_plane[base + i] = v[q];
  }
  if (TIMING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (TIMING && tid == 0) { atomicAdd(&g_time2[0], tl - t0); atomicAdd(&g_time2[1], wall_clock64() - tl); atomicAdd(&g_time2[4], 1ull); }
}

// device-scope load of one complex: always misses the L1 (TCP), served by the L2
__device__ __forceinline__ v2d load_sc1(const v2d* p) {
  const double* q = reinterpret_cast<const double*>(p);
  v2d r;
  r.x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}

template <bool NT_OUT, bool SC1 = false>
__device__ __forceinline__ void item_B(const Dims& d, const v2d* src_plane, v2d* out_plane, int item, int tid) {
  // tile of 8 columns: thread (t = tid % 8, jt = tid / 8) moves rows jt + 32 q
  const int t = tid % 8, jt = tid / 8;
  const int col = item * 8 + t;
  if (col >= d.nzc) return;
  v2d v[8];
  const unsigned long long t0 = wall_clock64();
  const int per = d.ny / 8;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = jt + q * 32;
    v[q] = (jt < 32 && j < d.ny) ? (SC1 ? load_sc1(&src_plane[(long)j * d.nzc + col]) : src_plane[(long)j * d.nzc + col]) : (v2d){0, 0};
  }
  (void)per;
  if (TIMING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long tl = wall_clock64();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = jt + q * 32;
    if (j < d.ny) {
      if (NT_OUT) __builtin_nontemporal_store(v[q], &out_plane[(long)j * d.nzc + col]);
      else out_plane[(long)j * d.nzc + col] = v[q];
    }
  }
  if (TIMING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (TIMING && tid == 0) { atomicAdd(&g_time2[2], tl - t0); atomicAdd(&g_time2[3], wall_clock64() - tl); atomicAdd(&g_time2[5], 1ull); }
}

__global__ __launch_bounds__(256) void k_two_A(Dims d, const v2d* in, v2d* tmp) {
  const int unit = blockIdx.x / d.nA, item = blockIdx.x % d.nA;
  item_A(d, in + (long)unit * d.plane, tmp + (long)unit * d.plane, item, threadIdx.x, true);
}
__global__ __launch_bounds__(256) void k_two_B(Dims d, const v2d* tmp, v2d* out) {
  const int unit = blockIdx.x / d.nB, item = blockIdx.x % d.nB;
  item_B<true>(d, tmp + (long)unit * d.plane, out + (long)unit * d.plane, item, threadIdx.x);
}

__device__ int g_abort[8];
__device__ unsigned long long g_time[8];   // wall_clock64 ticks (100 MHz): ticket, unit id, dependency wait, item A, item B, signal, count   // watchdog: which wait gave up (kind, xcd, sequence, value seen, wanted)

__device__ __forceinline__ int spin_until(int* p, int want, int kind, int x, long seq) {
  int v;
  long spins = 0;
  while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want) {
    __builtin_amdgcn_s_sleep(2);
    if (__hip_atomic_load(&g_abort[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return -1000000;
    if (++spins > 200000) {   // ~0.1 s: the probe must not hang the box
      if (atomicCAS(&g_abort[0], 0, kind) == 0) {
        g_abort[1] = x; g_abort[2] = (int)seq; g_abort[3] = v; g_abort[4] = want;
      }
      return -1000000;
    }
  }
  return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_fused(Dims d, Sync s, const v2d* in, v2d* out, v2d* scratch, int* xcd_hist, int* prog) {
  __shared__ long long sh[2];
  const int tid = threadIdx.x;
  const int x = xcc_id();
  if (tid == 0) atomicAdd(&xcd_hist[x], 1);   // (once per workgroup)
  const int per = d.nA + d.nB;
  v2d* my_scratch = scratch + (long)x * 2 * d.plane;
  for (;;) {
    // The hand-out runs on the whole first wave in uniform control flow (every lane polls the same address); only the
    // read-modify-writes are single-lane.  With the waits inside an `if (tid == 0)` the compiler's loop structurizer let
    // lanes 1..63 of that wave run ahead to the barrier while lane 0 was still waiting: the workgroup re-read the old ticket.
    unsigned long long c0 = wall_clock64(), c1 = 0, c2 = 0, c3 = 0;
    if (tid < 64) {
      unsigned lo = 0, hi = 0;
      if (tid == 0) {
        const unsigned long long t0 = __hip_atomic_fetch_add(&s.ticket[x], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo = (unsigned)t0;
        hi = (unsigned)(t0 >> 32);
      }
      lo = __builtin_amdgcn_readfirstlane(lo);
      hi = __builtin_amdgcn_readfirstlane(hi);
      const unsigned long long t = ((unsigned long long)hi << 32) | lo;
      c1 = wall_clock64();
      const long seq = (long)(t / per);
      const int i = (int)(t % per);
      int* uid = &s.unit_id[(long)x * s.units + seq];
      int u = 0;
      if (i == 0) {
        if (tid == 0) {
          u = (int)__hip_atomic_fetch_add(s.next_unit, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(uid, u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        u = __builtin_amdgcn_readfirstlane(u);
      } else {
        u = spin_until(uid, 1, 1, x, seq) - 1;
        if (u < 0) u = 1 << 30;
      }
      c2 = wall_clock64();
      if (u < s.units) {
        if (i < d.nA) {
          if (seq >= 2 && spin_until(&s.doneB[(long)x * s.units + seq - 2], d.nB, 2, x, seq) < 0) u = 1 << 30;
        } else {
          if (spin_until(&s.doneA[(long)x * s.units + seq], d.nA, 3, x, seq) < 0) u = 1 << 30;
        }
      }
      c3 = wall_clock64();
      if (tid == 0) {
        sh[0] = ((long long)seq << 32) | (unsigned)i;
        sh[1] = u;
        if (TIMING) {
          atomicAdd(&g_time[0], c1 - c0);
          atomicAdd(&g_time[1], c2 - c1);
          atomicAdd(&g_time[2], c3 - c2);
          atomicAdd(&g_time[6], 1ull);
        }
      }
    }
    __syncthreads();
    const long seq = (long)(sh[0] >> 32);
    const int i = (int)(sh[0] & 0xffffffff);
    const int u = (int)sh[1];
    __syncthreads();
    if (u >= s.units) {   // (progress in pinned host memory for the host-side watchdog: on exit only, a PCIe atomic costs 20 us)
      if (tid == 0) __hip_atomic_fetch_add(&prog[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    v2d* buf = my_scratch + (seq & 1) * d.plane;
    const unsigned long long c4 = wall_clock64();
    if (i < d.nA) {
      item_A(d, in + (long)u * d.plane, buf, i, tid, true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned long long c5 = wall_clock64();
      if (tid == 0) __hip_atomic_fetch_add(&s.doneA[(long)x * s.units + seq], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (TIMING && tid == 0) { atomicAdd(&g_time[3], c5 - c4); atomicAdd(&g_time[5], wall_clock64() - c5); }
    } else {
      if (MODE == 0) asm volatile("buffer_inv sc1" ::: "memory");
      if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
      if (MODE == 2) item_B<true, true>(d, buf, out + (long)u * d.plane, i - d.nA, tid);
      else item_B<true>(d, buf, out + (long)u * d.plane, i - d.nA, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned long long c5 = wall_clock64();
      if (tid == 0) __hip_atomic_fetch_add(&s.doneB[(long)x * s.units + seq], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (TIMING && tid == 0) { atomicAdd(&g_time[4], c5 - c4); atomicAdd(&g_time[5], wall_clock64() - c5); }
    }
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  const int wgs_per_cu = argc > 2 ? atoi(argv[2]) : 3;
  Dims d;
  d.nx = n; d.ny = n;
  d.nzc = ((n / 2 + 1 + 7) / 8) * 8;
  d.ncomp = 3;
  d.rows_per_item = 16;
  d.nA = d.ny / d.rows_per_item;
  d.nB = (d.nzc + 7) / 8;
  d.plane = (long)d.ny * d.nzc;
  const int units = d.nx * d.ncomp;
  const long total = (long)units * d.plane;
  printf("n %d: plane %.2f MB, units %d, field %.1f MB, nA %d nB %d\n", n, d.plane * 16 / 1e6, units, total * 16 / 1e6, d.nA, d.nB);
  v2d *in, *out, *tmp, *scratch;
  CK(hipMalloc(&in, total * 16)); CK(hipMalloc(&out, total * 16)); CK(hipMalloc(&tmp, total * 16));
  CK(hipMalloc(&scratch, 8 * 2 * d.plane * 16));
  std::vector<double> h(2 * total);
  for (long i = 0; i < 2 * total; ++i) h[i] = (double)(i % 1000003) * 0.5;
  CK(hipMemcpy(in, h.data(), total * 16, hipMemcpyHostToDevice));
  Sync s;
  s.units = units;
  char* pool;
  const size_t pool_bytes = 9 * sizeof(unsigned long long) + 3 * 8 * (size_t)units * sizeof(int) + 8 * sizeof(int);
  CK(hipMalloc(&pool, pool_bytes));
  s.ticket = (unsigned long long*)pool;
  s.next_unit = s.ticket + 8;
  s.unit_id = (int*)(s.next_unit + 1);
  s.doneA = s.unit_id + 8 * units;
  s.doneB = s.doneA + 8 * units;
  int* xcd_hist = s.doneB + 8 * units;
  int* prog;
  CK(hipHostMalloc(&prog, 64, hipHostMallocDefault));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int dev_cus = 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  dev_cus = prop.multiProcessorCount;
  const double bytes_min = 2.0 * total * 16;   // one read + one write
  auto check = [&](const char* what) {
    std::vector<double> r(2 * total);
    CK(hipMemcpy(r.data(), out, total * 16, hipMemcpyDeviceToHost));
    long bad = 0;
    for (long i = 0; i < 2 * total; ++i) bad += r[i] != h[i];
    printf("  %s: %ld wrong values\n", what, bad);
    CK(hipMemset(out, 0, total * 16));
  };
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) {
      hipLaunchKernelGGL(k_two_A, dim3(units * d.nA), dim3(256), 0, 0, d, in, tmp);
      hipLaunchKernelGGL(k_two_B, dim3(units * d.nB), dim3(256), 0, 0, d, tmp, out);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("two kernels : %.3f ms per pair, %.0f GB/s of the 2x-traffic (%.0f GB/s of the minimal bytes)\n", ms / 10,
           2 * bytes_min / (ms / 10) / 1e6, bytes_min / (ms / 10) / 1e6);
  }
  check("two kernels");
  {
    unsigned long long t2[8];
    CK(hipMemcpyFromSymbol(t2, HIP_SYMBOL(g_time2), sizeof(t2)));
    printf("  two kernels, inside the bodies (us): A loads %.2f, A stores %.2f | B loads %.2f, B stores %.2f\n", t2[0] / 100.0 / t2[4],
           t2[1] / 100.0 / t2[4], t2[2] / 100.0 / t2[5], t2[3] / 100.0 / t2[5]);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_time2), z, sizeof(z)));
  }
  auto run_fused = [&](int mode) {
  for (int rep = 0; rep < 2; ++rep) {
    float sum = 0;
    for (int it = 0; it < 10; ++it) {
      CK(hipMemsetAsync(pool, 0, pool_bytes, 0));
      CK(hipEventRecord(e0));
      prog[0] = prog[1] = 0;
      const dim3 grid(wgs_per_cu > 16 ? wgs_per_cu : dev_cus * wgs_per_cu);
      if (mode == 0) hipLaunchKernelGGL(k_fused<0>, grid, dim3(256), 0, 0, d, s, in, out, scratch, xcd_hist, prog);
      if (mode == 1) hipLaunchKernelGGL(k_fused<1>, grid, dim3(256), 0, 0, d, s, in, out, scratch, xcd_hist, prog);
      if (mode == 2) hipLaunchKernelGGL(k_fused<2>, grid, dim3(256), 0, 0, d, s, in, out, scratch, xcd_hist, prog);
      if (mode == 3) hipLaunchKernelGGL(k_fused<3>, grid, dim3(256), 0, 0, d, s, in, out, scratch, xcd_hist, prog);
      CK(hipEventRecord(e1));
      for (int w = 0; w < 100 && hipEventQuery(e1) != hipSuccess; ++w) {
        usleep(100000);
        if (w == 99) { printf("    STUCK: %d items started, %d workgroups left\n", ((volatile int*)prog)[0], ((volatile int*)prog)[1]); _exit(3); }
      }
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      sum += ms;
    }
    printf("fused via L2, mode %d (0 buffer_inv sc1, 1 buffer_inv sc0, 2 sc1 loads, 3 nothing; %d workgroups): %.3f ms per pair, %.0f GB/s of the minimal bytes\n",
           mode, wgs_per_cu > 16 ? wgs_per_cu : dev_cus * wgs_per_cu, sum / 10, bytes_min / (sum / 10) / 1e6);
  }
  check("fused");
  unsigned long long tm[8];
  CK(hipMemcpyFromSymbol(tm, HIP_SYMBOL(g_time), sizeof(tm)));
  const double per = 1.0 / (tm[6] ? tm[6] : 1) / 100.0;   // us per item (100 MHz)
  printf("  per item (us): ticket %.2f, unit id %.2f, dependency wait %.2f, A body %.2f (x nA/(nA+nB)), B body %.2f, signal %.2f\n", tm[0] * per,
         tm[1] * per, tm[2] * per, tm[3] * per, tm[4] * per, tm[5] * per);
  unsigned long long t2[8];
  CK(hipMemcpyFromSymbol(t2, HIP_SYMBOL(g_time2), sizeof(t2)));
  printf("  inside the bodies (us): A loads %.2f, A stores %.2f | B loads %.2f, B stores %.2f   (two-kernel launches before count too)\n",
         t2[0] / 100.0 / (t2[4] ? t2[4] : 1), t2[1] / 100.0 / (t2[4] ? t2[4] : 1), t2[2] / 100.0 / (t2[5] ? t2[5] : 1), t2[3] / 100.0 / (t2[5] ? t2[5] : 1));
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_time2), z, sizeof(z)));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_time), z, sizeof(z)));
  };
  run_fused(2);
  run_fused(0);
  int ab[8];
  CK(hipMemcpyFromSymbol(ab, HIP_SYMBOL(g_abort), sizeof(ab)));
  if (ab[0]) {
    printf("  WATCHDOG: wait kind %d (1 unit id, 2 doneB, 3 doneA) on xcd %d sequence %d saw %d, wanted %d\n", ab[0], ab[1], ab[2], ab[3], ab[4]);
    std::vector<int> st(3 * 8 * units);
    unsigned long long tk[9];
    CK(hipMemcpy(tk, pool, sizeof(tk), hipMemcpyDeviceToHost));
    CK(hipMemcpy(st.data(), s.unit_id, st.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int x = 0; x < 8; ++x) {
      printf("  xcd %d ticket %llu :", x, tk[x]);
      for (int q = 0; q < 6; ++q) printf(" [u %d A %d B %d]", st[x * units + q], st[8 * units + x * units + q], st[16 * units + x * units + q]);
      printf("\n");
    }
    printf("  next_unit %llu\n", tk[8]);
  }
  int hist[8];
  CK(hipMemcpy(hist, xcd_hist, sizeof(hist), hipMemcpyDeviceToHost));
  printf("  workgroups per XCC id (last launch): %d %d %d %d %d %d %d %d\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
  check("fused");
  return 0;
}
