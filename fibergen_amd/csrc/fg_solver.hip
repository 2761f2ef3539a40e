#include "fg_solver.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

#include "fg_hip_util.h"
#include "fg_slab.h"

namespace fg {

using namespace hostmath;

using namespace slots;

namespace {
constexpr double kEps = 2.220446049250313e-16;

// The stop rule's hand-over to the host: the reduced sums (a block of dscal_) and the error flag are written straight into
// pinned host memory, followed by a sequence number the host spins on -- one tiny kernel instead of two or three small
// copies and an event wait (128^3: 0.37 -> 0.2x ms per pass of fg_run_load_case, the copies sat on the stream's critical path).
__global__ void k_publish(const double* dsc, int n, const double* dsc2, int n2, const int* derr, double* hsc, double* hsc2,
                          int* herr, volatile unsigned* hseq, unsigned seq) {
  const int t = threadIdx.x;
  if (t < n) hsc[t] = dsc[t];
  if (t < n2) hsc2[t] = dsc2[t];
  if (t == 0) *herr = *derr;
  __threadfence_system();
  __syncthreads();
  if (t == 0) *hseq = seq;
}

// Viscosity mode with mixed boundary conditions: DeltaOperatorStaggered F:20422-20460 hands adj = E - 2 alpha m <tau> to
// GammaOperatorStaggered, whose applyBCProjector (F:20263-20270, bc_relax = 1) adds alpha MQ:<tau> at the end.  The tail sweep
// forms  adj_c = E_c - coef * sums_c / N  from the six sums the divergence sweep left on the device (coef = 2 alpha m); this
// kernel replaces the sums by  s - (alpha / coef) B s,  B the plain 6 x 6 matrix of v -> MQ:v, so that the sweep's adj carries
// both terms -- no host round trip.
struct Mat36 {
  double a[36];
};
__global__ void k_bc_adjust_sums(double* sums, Mat36 B, double factor) {
  if (threadIdx.x != 0) return;
  double s[6], o[6];
  for (int c = 0; c < 6; ++c) s[c] = sums[c];
  for (int c = 0; c < 6; ++c) {
    double t = 0.0;
    for (int j = 0; j < 6; ++j) t += B.a[c * 6 + j] * s[j];
    o[c] = s[c] - factor * t;
  }
  for (int c = 0; c < 6; ++c) sums[c] = o[c];
}

double now_seconds() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}
}  // namespace

Solver::Solver(int nx, int ny, int nz, double dx, double dy, double dz, int device, int rank, int nranks, bool slab_layout,
               hipStream_t shared_stream)
    : slab_layout_(slab_layout), device_(device), rank_(rank), nranks_(nranks), nxg_(nx) {
  if (nx < 1 || ny < 1 || nz < 1) throw std::runtime_error("grid dimensions must be >= 1");
  if (!(dx > 0) || !(dy > 0) || !(dz > 0)) throw std::runtime_error("RVE dimensions must be > 0");
  if (nranks < 1 || rank < 0 || rank >= nranks) throw std::runtime_error("invalid rank / number of ranks");
  if (nranks > 1 && (nx % nranks != 0 || ny % nranks != 0))
    throw std::runtime_error("slab decomposition needs nx and ny divisible by the number of ranks");
  // x-slab: this rank owns planes [rank*nxl, (rank+1)*nxl) of every component; cell sizes stay global
  const int nxl = nx / nranks;
  nyl_ = ny / nranks;
  g_ = make_grid(nxl, ny, nz, dx, dy, dz);
  g_.hx = nx / dx;
  nglobal_ = (long)nx * ny * nz;
  if (g_.n >= (1L << 31)) throw std::runtime_error("grid too large: padded component exceeds 2^31 reals");
  int ndev = 0;
  FG_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (ndev < 1) throw std::runtime_error("no HIP device available: fibergen_amd needs an AMD GPU (gfx950)");
  if (device < 0 || device >= ndev) throw std::runtime_error("invalid device index");
  FG_HIP_CHECK(hipSetDevice(device));
  try {   // a failure half way (out of memory at 512^3, say) must give everything back: the destructor will not run
  if (shared_stream) {
    stream_ = shared_stream;
    owns_stream_ = false;
  } else {
    FG_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  }
  comm_stream_ = stream_;
  FG_HIP_CHECK(hipEventCreate(&ev_[0]));
  FG_HIP_CHECK(hipEventCreate(&ev_[1]));
  FG_HIP_CHECK(hipEventCreateWithFlags(&ev_copy_, hipEventDisableTiming));
  pt_.n = 0;
  for (int i = 0; i < kMaxPhases; ++i) pt_.mu[i] = pt_.lambda[i] = 0.0;
  opt_.mu_0 = std::numeric_limits<double>::quiet_NaN();  // F:15340
  opt_.eps_a = std::pow(kEps, 2.0 / 3.0);

  const size_t comp = g_.n * sizeof(double);
  FG_HIP_CHECK(hipMalloc(&eps_, 6 * comp));
  FG_HIP_CHECK(hipMalloc(&tau_, 6 * comp));
  FG_HIP_CHECK(hipMalloc(&fu_, 3 * comp));
  FG_HIP_CHECK(hipMalloc(&fu_alt_, 3 * comp));
  FG_HIP_CHECK(hipMemsetAsync(fu_alt_, 0, 3 * comp, stream_));
  FG_HIP_CHECK(hipMemsetAsync(eps_, 0, 6 * comp, stream_));
  FG_HIP_CHECK(hipMemsetAsync(tau_, 0, 6 * comp, stream_));
  FG_HIP_CHECK(hipMemsetAsync(fu_, 0, 3 * comp, stream_));
  {
    const long rows = std::max<long>(partial_rows(g_), kMaxReduceBlocks);
    FG_HIP_CHECK(hipMalloc(&partial_, (size_t)rows * 8 * sizeof(double)));
  }
  FG_HIP_CHECK(hipMalloc(&dscal_, kNumSlots * sizeof(double)));
  FG_HIP_CHECK(hipHostMalloc(&hscal_, kNumSlots * sizeof(double)));
  FG_HIP_CHECK(hipMalloc(&derr_, 2 * sizeof(int)));   // [0] kernel error flag, [1] scratch of two_phase_complementary
  FG_HIP_CHECK(hipHostMalloc(&herr_, 2 * sizeof(int)));   // [0] error flag, [1] sequence number of k_publish
  herr_[1] = 0;
  hseq_ = reinterpret_cast<unsigned*>(herr_ + 1);
  FG_HIP_CHECK(hipMemsetAsync(derr_, 0, sizeof(int), stream_));
  // fields of this size come back through the staged pipeline: its pinned buffers (a one-time cost of the process, ~18 ms)
  // are allocated here rather than inside the first fg_get_field
  if (staged_copy(6 * comp, true)) (void)HostStager::of_device(device_);

  fft_.reset(new Fft3(g_, stream_));
  if (slab_layout_) {   // halo planes of the strain-state pipeline of the slab driver (tau going out, tau coming in)
    const size_t plane = (size_t)g_.nyzp * sizeof(double);
    for (int k = 0; k < 4; ++k) {
      FG_HIP_CHECK(hipMalloc(&halo_[k], 2 * plane));
      FG_HIP_CHECK(hipMemsetAsync(halo_[k], 0, 2 * plane, stream_));
    }
  }

  // Separable factors of G0OperatorFourierStaggeredGeneral  F:19838-19876, evaluated on the
  // host with the same libm calls the reference makes per frequency.
  const int len[3] = {nxg_, ny, nz};
  const double d[3] = {dx, dy, dz};
  for (int a = 0; a < 3; ++a) {
    const int n = len[a];
    const int cnt = (a == 2) ? g_.nzc : n;
    std::vector<double> kpm(cnt);
    std::vector<cplx> kp(cnt);
    const double h = d[a] / (2 * n);
    const double xi_0 = 2 * M_PI * h / d[a];
    const bool even = (n & 1) == 0;
    const size_t half = even ? (size_t)(n / 2 - 1) : (size_t)(n / 2);
    for (size_t i = 0; i < (size_t)cnt; ++i) {
      const double xi = xi_0 * ((i <= half) ? (double)i : ((double)i - (double)n));
      const double s = std::sin(xi) / h;
      const std::complex<double> z = s * std::exp(std::complex<double>(0, xi));
      kpm[i] = s;
      kp[i] = cmake(z.real(), z.imag());
    }
    FG_HIP_CHECK(hipMalloc(&g0_kpm_[a], cnt * sizeof(double)));
    FG_HIP_CHECK(hipMalloc(&g0_kp_[a], cnt * sizeof(cplx)));
    FG_HIP_CHECK(hipMemcpy(g0_kpm_[a], kpm.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(g0_kp_[a], kp.data(), cnt * sizeof(cplx), hipMemcpyHostToDevice));
    // GammaOperatorFourierCollocated  F:19385, 19411-19424: xi = (1/d) * signed index
    std::vector<double> xi(cnt);
    const double xi_c0 = 1 / d[a];
    for (size_t i = 0; i < (size_t)cnt; ++i) xi[i] = xi_c0 * ((i <= half) ? (double)i : ((double)i - (double)n));
    FG_HIP_CHECK(hipMalloc(&xi_[a], cnt * sizeof(double)));
    FG_HIP_CHECK(hipMemcpy(xi_[a], xi.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
  }

  BC_P_ = voigt_id4();
  for (int i = 0; i < 6; ++i) F00_[i] = 0.0, sumsq_[i] = 0.0;
  recompute_bc();
  reset_stage_times();
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  } catch (...) {
    release();
    throw;
  }
}

Solver::~Solver() {
  (void)hipSetDevice(device_);
  if (group_) group_->invalidate();
  release();
}

// frees every device / host resource the object holds (idempotent; also the constructor's failure path)
void Solver::release() {
  if (!stream_) return;
  (void)hipStreamSynchronize(stream_);
  if (comm_stream_ && comm_stream_ != stream_) (void)hipStreamSynchronize(comm_stream_);
  fft_.reset();
  for (int k = 0; k < 4; ++k)
    if (halo_[k]) (void)hipFree(halo_[k]);
  for (void* p : {(void*)lam_phic_, (void*)lam_nrmc_, (void*)lam_epsc_})
    if (p) (void)hipFree(p);
  if (mixed_list_) (void)hipFree(mixed_list_);
  if (aff_list_) (void)hipFree(aff_list_);
  if (aff_slots_) (void)hipFree(aff_slots_);
  if (dtau_) (void)hipFree(dtau_);
  double* bufs[] = {eps_, tau_, fu_, fu_alt_, phi_, normals_, partial_, dscal_, cg_r_, cg_p_, cg_w_, mod_, fu_cg_};
  for (double* b : bufs)
    if (b) (void)hipFree(b);
  if (hscal_) (void)hipHostFree(hscal_);
  if (derr_) (void)hipFree(derr_);
  if (herr_) (void)hipHostFree(herr_);
  for (int a = 0; a < 3; ++a) {
    if (g0_kpm_[a]) (void)hipFree(g0_kpm_[a]);
    if (g0_kp_[a]) (void)hipFree(g0_kp_[a]);
    if (xi_[a]) (void)hipFree(xi_[a]);
  }
  for (hipEvent_t e : {ev_[0], ev_[1], ev_copy_})
    if (e) (void)hipEventDestroy(e);
  if (aux_stream_) {
    (void)hipStreamSynchronize(aux_stream_);
    (void)hipStreamDestroy(aux_stream_);
    (void)hipEventDestroy(ev_fork_);
    (void)hipEventDestroy(ev_join_);
  }
  comm_.reset();
  fft_ys_.reset();
  for (double* b : {su_[0], su_[1], smod_, scg_, scg2_, su_alt_})
    if (b) (void)hipFree(b);
  if (ev_c2x_) (void)hipEventDestroy(ev_c2x_);
  if (ev_norm_) (void)hipEventDestroy(ev_norm_);
  for (hipEvent_t e : ev_ct_)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ev_x_)
    if (e) (void)hipEventDestroy(e);
  if (owns_comm_stream_ && comm_stream_) (void)hipStreamDestroy(comm_stream_);
  if (owns_stream_) (void)hipStreamDestroy(stream_);
  stream_ = comm_stream_ = nullptr;
}

// ------------------------------------------------------------------ configuration
void Solver::set_num_phases(int n) {
  if (n < 1 || n > kMaxPhases) throw std::runtime_error("number of phases must be in [1, 8]");
  FG_HIP_CHECK(hipSetDevice(device_));
  if (phi_) FG_HIP_CHECK(hipFree(phi_));
  phi_ = nullptr;
  FG_HIP_CHECK(hipMalloc(&phi_, (size_t)n * g_.n * sizeof(double)));
  FG_HIP_CHECK(hipMemsetAsync(phi_, 0, (size_t)n * g_.n * sizeof(double), stream_));
  pt_.n = n;
  mod_dirty_ = true;
  smod_dirty_ = true;
  complement_dirty_ = true;
  mixed_dirty_ = true;
}

void Solver::set_phase_material(int p, double mu, double lambda) {
  if (p < 0 || p >= pt_.n) throw std::runtime_error("phase index out of range");
  mod_dirty_ = true;
  smod_dirty_ = true;
  complement_dirty_ = true;
  mixed_dirty_ = true;
  pt_.mu[p] = mu;
  pt_.lambda[p] = lambda;
}

void Solver::set_phase_field(int p, const double* phi_host) {
  if (p < 0 || p >= pt_.n) throw std::runtime_error("phase index out of range");
  mod_dirty_ = true;
  smod_dirty_ = true;
  complement_dirty_ = true;
  mixed_dirty_ = true;
  upload_padded(phi_ + (long)p * g_.n, phi_host);
}

void Solver::set_normals(const double* n_host) {
  FG_HIP_CHECK(hipSetDevice(device_));
  mixed_dirty_ = true;   // the interface lists carry compact copies of the normals
  if (!normals_) {
    FG_HIP_CHECK(hipMalloc(&normals_, 3 * g_.n * sizeof(double)));
    FG_HIP_CHECK(hipMemsetAsync(normals_, 0, 3 * g_.n * sizeof(double), stream_));
  }
  upload_padded(normals_, n_host, 3, g_.n);
}

void Solver::set_bc_projector(const double* P36) {
  // setBCProjector checks  F:20601-20615
  Mat6 P;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) P.a[i][j] = P36[i * 6 + j];
  const double se = std::sqrt(kEps);
  Mat6 D = P;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) D.a[i][j] = P.a[i][j] - P.a[j][i];
  if (frobenius(D) > se) throw std::runtime_error("Projector is not symmetric");
  Mat6 PP = voigt_mm(P, P);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) D.a[i][j] = P.a[i][j] - PP.a[i][j];
  if (frobenius(D) > se) throw std::runtime_error("Specified Projector is not a projector");
  BC_P_ = P;
  recompute_bc();
}

// setBCProjector body  F:20617-20664 (depends on the current reference material)
void Solver::recompute_bc() {
  const Mat6 Id = voigt_id4(), II = voigt_ii4();
  bool q_zero = true;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      BC_Q_.a[i][j] = Id.a[i][j] - BC_P_.a[i][j];
      if (BC_Q_.a[i][j] != 0.0) q_zero = false;
    }
  if (q_zero) {  // pure strain BC: Q:C0, M, MQ vanish whatever C0 is (also while mu_0 is NaN)
    BC_QC0_ = BC_M_ = BC_MQ_ = mat6_zero();
    return;
  }
  Mat6 C0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) C0.a[i][j] = 2 * opt_.mu_0 * Id.a[i][j] + opt_.lambda_0 * II.a[i][j];
  BC_QC0_ = voigt_mm(BC_Q_, C0);
  if (std::isnan(opt_.mu_0)) {
    // run() calls setBCProjector before calcRefMaterial replaced the NaN (F:21354 vs F:21742);
    // the values are recomputed before their first use.
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) BC_M_.a[i][j] = BC_MQ_.a[i][j] = std::numeric_limits<double>::quiet_NaN();
    return;
  }
  BC_M_ = bc_pseudo_inverse(voigt_mm(BC_QC0_, BC_Q_));
  BC_MQ_ = voigt_mm(BC_M_, BC_Q_);
}

// ------------------------------------------------------------------ helpers
FieldPtrs<6> Solver::ptrs6(double* base) const {
  FieldPtrs<6> f;
  for (int c = 0; c < 6; ++c) f.p[c] = base + (long)c * g_.n;
  return f;
}
FieldPtrs<3> Solver::ptrs3(double* base) const {
  FieldPtrs<3> f;
  for (int c = 0; c < 3; ++c) f.p[c] = base + (long)c * g_.n;
  return f;
}

// mode = viscosity: ScalarLinearIsotropicMaterialLaw(6) with mu *= 0.5 (F:15234-15239), S = E * (alpha * mu / 2).
// The Hooke kernels give S = E * (2 alpha mu') + alpha lambda' tr(E): mu' = mu / 4, lambda' = 0 is the same
// arithmetic (scalings by powers of two are exact, the added zero changes nothing).
PhaseTable Solver::phase_table() const {
  PhaseTable t = pt_;
  if (opt_.mode == 2)
    for (int q = 0; q < kMaxPhases; ++q) t.mu[q] = 0.25 * pt_.mu[q], t.lambda[q] = 0.0;
  return t;
}

StressParams Solver::stress_params(double mu_0, double lambda_0, double alpha) const {
  StressParams sp;
  sp.pt = phase_table();
  sp.mixing = opt_.mixing;
  sp.mu_0 = mu_0;
  sp.lambda_0 = lambda_0;
  sp.alpha = alpha;
  sp.eps_g = opt_.eps_g;
  sp.eps_a = opt_.eps_a;
  return sp;
}

// Boundary transfers (GetField / SetField F:26931-27010 copy row by row): large fields go through the staged pipeline of
// fg_transfer.h (padding stripped / added on the device, whole chunks over the link, a team of host threads on the pageable
// side), small ones through one strided copy.
// Measured on the MI355X boxes (tools/transfer_probe.py, 805 MB): uploads from pageable memory already run at the link's
// rate through the runtime's own staging (54 GB/s), downloads into FRESH pageable memory do not (48 ms: the page faults of
// the destination are taken one by one by the copying thread) -- the pipeline takes them on its team of host threads while
// the next chunks are on the link: 18-19 ms (14.2 ms into memory that is already mapped, either way).
bool Solver::staged_copy(size_t bytes, bool download) const {
  return opt_.staged_copy > 0 || (opt_.staged_copy < 0 && download && bytes >= (size_t)8 << 20);
}

void Solver::upload_rows(const std::vector<RowBlock>& blocks, long len, long pitch) {
  FG_HIP_CHECK(hipSetDevice(device_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  size_t bytes = 0;
  for (const RowBlock& b : blocks) bytes += (size_t)b.nrows * len * sizeof(double);
  if (staged_copy(bytes, false)) {
    HostStager::of_device(device_).upload(blocks, len, pitch, (size_t)std::max(1, opt_.stage_chunk_kb) << 10);
    return;
  }
  for (const RowBlock& b : blocks)
    FG_HIP_CHECK(hipMemcpy2D(b.dev, pitch * sizeof(double), b.host, len * sizeof(double), len * sizeof(double), (size_t)b.nrows,
                             hipMemcpyHostToDevice));
}

void Solver::download_rows(const std::vector<RowBlock>& blocks, long len, long pitch) {
  FG_HIP_CHECK(hipSetDevice(device_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  size_t bytes = 0;
  for (const RowBlock& b : blocks) bytes += (size_t)b.nrows * len * sizeof(double);
  if (staged_copy(bytes, true)) {
    HostStager::of_device(device_).download(blocks, len, pitch, (size_t)std::max(1, opt_.stage_chunk_kb) << 10);
    return;
  }
  for (const RowBlock& b : blocks)
    FG_HIP_CHECK(hipMemcpy2D(b.host, len * sizeof(double), b.dev, pitch * sizeof(double), len * sizeof(double), (size_t)b.nrows,
                             hipMemcpyDeviceToHost));
}

// nc padded components [nx][ny][nzp], `dstride` doubles apart on the device <-> nc unpadded host components
void Solver::upload_padded(double* dst, const double* src, int nc, long dstride) {
  std::vector<RowBlock> blocks;
  for (int c = 0; c < nc; ++c)
    blocks.push_back(RowBlock{dst + (long)c * dstride, const_cast<double*>(src) + (long)c * g_.nxyz, (long)g_.nx * g_.ny});
  upload_rows(blocks, g_.nz, g_.nzp);
}

void Solver::download_unpadded(const double* src, double* dst, int nc, long dstride) {
  std::vector<RowBlock> blocks;
  for (int c = 0; c < nc; ++c)
    blocks.push_back(RowBlock{const_cast<double*>(src) + (long)c * dstride, dst + (long)c * g_.nxyz, (long)g_.nx * g_.ny});
  download_rows(blocks, g_.nz, g_.nzp);
}

void Solver::check_device_error(const char* where) {
  FG_HIP_CHECK(hipMemcpyAsync(herr_, derr_, sizeof(int), hipMemcpyDeviceToHost, stream_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  if (*herr_ != 0) {
    FG_HIP_CHECK(hipMemsetAsync(derr_, 0, sizeof(int), stream_));
    if (opt_.mixing == kMixLaminate)
      throw std::runtime_error(std::string("The laminate mixing rule supports only two phase mixtures (") + where + ")");
    throw std::runtime_error(std::string("device kernel reported an error (") + where + ")");
  }
}

// The same check, but the host waits only for the copies enqueued so far (an event), not for work enqueued after them.
void Solver::fetch_norms_and_errors(const char* where) {
  static_assert(kSlotMean == kSlotSumSq + 6, "the displacement sweep writes norms and tau sums as one block of 12");
  const bool mixed_u = pending_back_ && !(frobenius(BC_MQ_) < kEps);
  const int nfetch = mixed_u ? 12 : 6;
  const int n2 = (mixed_u && opt_.mixing != kMixVoigt) ? 6 : 0;   // mixed BC + laminate: + sums of the interface differences
  {
    // one tiny kernel publishes sums + error flag + sequence number in pinned host memory; the host spins on the number
    const unsigned seq = ++publish_seq_;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, stream_, dscal_ + kSlotSumSq, nfetch, dscal_ + kSlotScratch, n2, derr_,
                       hscal_ + kSlotSumSq, hscal_ + kSlotScratch, herr_, hseq_, seq);
    FG_HIP_CHECK(hipGetLastError());
    if (pending_back_) launch_pending_back();   // speculative: u_{k+1} is built while the host looks at the norms of eps_k
    long spins = 0;
    while (__atomic_load_n(hseq_, __ATOMIC_ACQUIRE) != seq) {
      if (++spins > (1L << 22)) {   // a long pass (or a stuck device): fall back to a blocking wait once in a while
        FG_HIP_CHECK(hipStreamQuery(stream_) == hipErrorNotReady ? hipSuccess : hipStreamSynchronize(stream_));
        spins = 0;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  if (*herr_ != 0) {
    FG_HIP_CHECK(hipMemsetAsync(derr_, 0, sizeof(int), stream_));
    if (opt_.mixing == kMixLaminate)
      throw std::runtime_error(std::string("The laminate mixing rule supports only two phase mixtures (") + where + ")");
    throw std::runtime_error(std::string("device kernel reported an error (") + where + ")");
  }
}

// FFT / Green-operator chain of the displacement loop, enqueued without touching the host-side state: fu_alt_ then holds
// u_{k+1}; adopt_back() makes it the current state.  If the iteration stops first, it is simply never adopted.
void Solver::launch_pending_back() {
  fft_g0_chain(fu_alt_, -1.0, nullptr, opt_.mode == 0 ? tau_ : nullptr);   // tau_ is free in the displacement loop
  pending_back_ = false;
  back_ready_ = true;
}

void Solver::adopt_back() {
  if (!back_ready_) launch_pending_back();
  back_ready_ = false;
  double* t = fu_;
  fu_ = fu_alt_;
  fu_alt_ = t;
  u_valid_ = true;
  eps_stale_ = true;
  for (int c = 0; c < 6; ++c) E_cur_[c] = E_next_[c];
  if (timing_) times_.count++;
}

void Solver::enable_stage_timing(bool on) {
  if (on && !timing_) {
    reset_stage_times();   // a new measurement starts from zero
    // what a pair of events reads with NOTHING between them (the smallest of 16 pairs on the idle stream): every
    // measurement carries it on top of the kernel's duration, so it is subtracted (rocprofv3's kernel durations then agree)
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    double bias = 1e30;
    for (int i = 0; i < 16; ++i) {
      FG_HIP_CHECK(hipEventRecord(ev_[0], stream_));
      FG_HIP_CHECK(hipEventRecord(ev_[1], stream_));
      FG_HIP_CHECK(hipEventSynchronize(ev_[1]));
      float ms = 0.f;
      FG_HIP_CHECK(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
      if (ms < bias) bias = ms;
    }
    event_bias_ms_ = bias;
  }
  timing_ = on;
}
long Solver::counter(const std::string& name) const {
  if (name == "interface_voxels") return (long)mixed_n_;
  if (name == "affected_voxels") return (long)aff_n_;
  if (name == "pair_chunk_planes") return (long)pair_chunk_planes(opt_.mode == 1 ? 1 : 3);
  return -1;
}

void Solver::reset_stage_times() {
  for (int i = 0; i < kNumTimedKernels; ++i) times_.ms[i] = 0.0;
  for (int i = 0; i < 4; ++i) comm_ms_[i] = 0.0;
  times_.count = 0;
}
void Solver::time_begin(int) {
  if (timing_) FG_HIP_CHECK(hipEventRecord(ev_[0], stream_));
}
void Solver::time_end(int stage) {
  if (!timing_) return;
  FG_HIP_CHECK(hipEventRecord(ev_[1], stream_));
  FG_HIP_CHECK(hipEventSynchronize(ev_[1]));
  float ms = 0.f;
  FG_HIP_CHECK(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
  times_.ms[stage] += ms > event_bias_ms_ ? ms - event_bias_ms_ : 0.0;
}

// ------------------------------------------------------------------ one pass of the basic scheme
// basicScheme  F:20558-20578 + GammaOperatorStaggered  F:20288-20300:
//   tau = (C - C0):eps ; f = div tau ; u = G0 f ; eps = E + sym grad u (+ R)
void Solver::basic_scheme(const double* E6, double* src, double* dst) {
  if (!src) src = eps_;
  if (!dst) dst = eps_;
  if (nranks_ != 1) throw std::runtime_error("basic_scheme: slab solvers run under the slab driver (fg_slab.hip)");
  if (pt_.n < 1) throw std::runtime_error("No materials specified");
  if (opt_.mixing == kMixLaminate && !normals_) throw std::runtime_error("laminate mixing needs interface normals");
  FieldPtrs<kMaxPhases> phi;
  for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
  FieldPtrs<3> nrm;
  for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
  const double alpha = -1.0;  // GammaOperator(..., -1)  F:20575

  ensure_eps();  // the displacement-based loop may have left the strain implicit
  if (opt_.bc_relax != 1.0) {  // F:20563-20565: mean of the operator's argument
    launch_sum6(g_, ptrs6(src), false, partial_, dscal_ + kSlotMean, stream_);
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean, dscal_ + kSlotMean, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    for (int c = 0; c < 6; ++c) F00_[c] = hscal_[kSlotMean + c] / (double)nglobal_;
  }

  if (opt_.mode == 2) {
    // DeltaOperatorStaggered  F:20422-20460 (dual Stokes scheme), called with alpha = -1 by basicScheme:
    //   m = 1/(4 mu0);  adj = E - 2 alpha m <tau>;  eta = GammaStaggered(adj; mu = -1/(4 m), lambda = inf)(tau)
    //   + 2 alpha m tau.   lambda0 = inf makes c20 = c10 in G0OperatorFourierStaggered (F:19749-19755).
    if (opt_.gamma_scheme != 0) throw std::runtime_error("viscosity mode supports gamma_scheme=staggered only");
    if (opt_.mixing != kMixVoigt) throw std::runtime_error("viscosity mode supports Voigt mixing only");
    if (opt_.bc_relax != 1.0) throw std::runtime_error("viscosity mode supports bc_relax = 1 only");
    const bool mixed_bc = !(frobenius(BC_MQ_) < kEps);   // initBCProjector / applyBCProjector inside GammaOperatorStaggered
    const double m = 1 / (4 * opt_.mu_0);
    const bool fused = opt_.fuse_stress_div != 0;
    if (fused) {
      // the polarisation is a point-wise function of the strain: it is evaluated inside the divergence sweep (with its
      // six sums) and again in the tail sweep, and never stored
      time_begin(0);
      if (opt_.u_loop >= 2 && u_tile_supported(g_))   // fast kernels allowed: the LDS-tiled marching form
        launch_eps_tile(g_, opt_.mu_0, opt_.lambda_0, ptrs6(src), effective_moduli(), ptrs3(fu_), partial_,
                        dscal_ + kSlotMean, stream_);
      else
        launch_stress_div_sum_voigt(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(src), phi, ptrs3(fu_), partial_,
                                    dscal_ + kSlotMean, stream_);
      time_end(0);
    } else {
      time_begin(0);
      launch_stress(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(src), phi, nrm, ptrs6(tau_), derr_, stream_);
      time_end(0);
      launch_sum6(g_, ptrs6(tau_), false, partial_, dscal_ + kSlotMean, stream_);   // tau_copy->average(), stays on the device
      time_begin(1);
      launch_div(g_, ptrs6(tau_), ptrs3(fu_), XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
      time_end(1);
    }
    if (mixed_bc) {
      Mat36 B;
      for (int j = 0; j < 6; ++j) {
        double e[6] = {0, 0, 0, 0, 0, 0}, col[6];
        e[j] = 1.0;
        voigt_mv(BC_MQ_, e, col);
        for (int c = 0; c < 6; ++c) B.a[c * 6 + j] = col[c];
      }
      hipLaunchKernelGGL(k_bc_adjust_sums, dim3(1), dim3(64), 0, stream_, dscal_ + kSlotMean, B, alpha / (2 * alpha * m));
      FG_HIP_CHECK(hipGetLastError());
    }
    const double mu_g = -1.0 / (4 * m);
    const double c12[2] = {-alpha / mu_g, -alpha / mu_g};
    fft_g0_chain(fu_, alpha, c12, fused ? tau_ : nullptr);   // (fused: the polarisation is never stored, its field is free for the x-contiguous layout)
    // adj = E - 2 alpha m <tau>;  eta = adj + sym grad u;  eta.xpay(eta, 2 alpha m, tau_copy)  F:20438-20452, one sweep
    Vec6 Ev;
    for (int c = 0; c < 6; ++c) Ev.v[c] = E6[c];
    time_begin(9);
    if (fused)
      launch_eps_delta_recompute(g_, ptrs3(fu_), ptrs6(src), stress_params(opt_.mu_0, opt_.lambda_0, 1.0), phi,
                                 dscal_ + kSlotMean, (double)nglobal_, Ev, 2 * alpha * m, ptrs6(dst), partial_,
                                 dscal_ + kSlotSumSq, stream_);
    else
      launch_eps_delta(g_, ptrs3(fu_), ptrs6(tau_), dscal_ + kSlotMean, (double)nglobal_, Ev, 2 * alpha * m, ptrs6(dst),
                       partial_, dscal_ + kSlotSumSq, stream_);
    time_end(9);
    if (timing_) times_.count++;
    u_valid_ = false;
    if (dst == eps_) eps_stale_ = false;
    for (int c = 0; c < 6; ++c) E_cur_[c] = E6[c];
    return;
  }
  if (opt_.gamma_scheme == 1) {
    // GammaOperatorCollocated  F:20302-20310: fftTensor, initBCProjector, Gamma0_hat, applyBCProjector, fftInvTensor on the
    // six components
    time_begin(0);
    launch_stress(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(src), phi, nrm, ptrs6(tau_), derr_, stream_);
    time_end(0);
    time_begin(2);
    fft_->forward(tau_, 6, g_.n, 1 / (double)nglobal_);   // fftTensor: 1/N on the forward transform  F:18531-18560
    time_end(2);
    XiTables xt;
    for (int a = 0; a < 3; ++a) xt.xi[a] = xi_[a];
    Vec6 Ev;
    for (int c = 0; c < 6; ++c) Ev.v[c] = E6[c];
    if (!(frobenius(BC_MQ_) < kEps) || opt_.bc_relax != 1.0) {
      // mixed boundary conditions: F0 = Re tau_hat(0) (initBCProjector(tau_hat) F:20219-20225; the forward transform is
      // scaled, so this is the mean of tau), and the zero frequency of eta_hat becomes E + R (applyBCProjector(eta_hat,
      // alpha) F:20272-20279) -- the Fourier kernel sets it to the value handed in
      double F0[6], t1[6], t2[6], t3[6];
      for (int c = 0; c < 6; ++c)
        FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean + c, tau_ + (long)c * g_.n, sizeof(double), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipStreamSynchronize(stream_));
      for (int c = 0; c < 6; ++c) F0[c] = hscal_[kSlotMean + c];
      voigt_mv(BC_MQ_, F0, t1);
      voigt_mv(BC_QC0_, F00_, t2);
      voigt_mv(BC_M_, t2, t3);
      for (int c = 0; c < 6; ++c) Ev.v[c] += alpha * (opt_.bc_relax * t1[c] - (1 - opt_.bc_relax) * t3[c]);
    }
    const double c10 = alpha / (4 * opt_.mu_0);
    const double c20 = -alpha / (opt_.mu_0 * (1 + opt_.mu_0 / (opt_.lambda_0 + opt_.mu_0)));
    time_begin(5);
    launch_gamma_collocated(g_, ptrs6(tau_), xt, c10, c20, 0.0, Ev, stream_);
    time_end(5);
    time_begin(8);
    fft_->inverse(tau_, 6, g_.n);
    time_end(8);
    time_begin(9);
    launch_copy(tau_, dst, 6 * g_.n, stream_);
    launch_sum6(g_, ptrs6(dst), true, partial_, dscal_ + kSlotSumSq, stream_);
    time_end(9);
    if (timing_) times_.count++;
    u_valid_ = false;
    if (dst == eps_) eps_stale_ = false;
    for (int c = 0; c < 6; ++c) E_cur_[c] = E6[c];
    return;
  }

  // initBCProjector  F:20228-20239 needs <tau> only for mixed boundary conditions
  double F0[6] = {0, 0, 0, 0, 0, 0};
  const bool mq_zero = frobenius(BC_MQ_) < kEps;
  // Voigt mixing: polarisation and divergence in one sweep (tau never stored); the laminate rule keeps the
  // two-kernel form (its per-voxel Newton solve is too costly to repeat at the six neighbours)
  const bool fuse_sd = opt_.fuse_stress_div && opt_.mixing == kMixVoigt && mq_zero;
  // with the fast kernels allowed (u_loop = 2) the LDS-tiled form takes over where the grid fits; it also delivers
  // the sums of tau, so mixed boundary conditions keep the fused sweep
  const bool tile_sd = opt_.fuse_stress_div && opt_.mixing == kMixVoigt && opt_.u_loop >= 2 && u_tile_supported(g_);
  if (tile_sd) {
    time_begin(0);
    launch_eps_tile(g_, opt_.mu_0, opt_.lambda_0, ptrs6(src), effective_moduli(), ptrs3(fu_), partial_, dscal_ + kSlotMean,
                    stream_);
    time_end(0);
    if (!mq_zero) {
      FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean, dscal_ + kSlotMean, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipStreamSynchronize(stream_));
      for (int c = 0; c < 6; ++c) F0[c] = hscal_[kSlotMean + c] / (double)nglobal_;
    }
  } else if (fuse_sd) {
    time_begin(0);
    launch_stress_div_voigt(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(src), phi, ptrs3(fu_), stream_);
    time_end(0);
  } else {
    time_begin(0);
    launch_stress(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(src), phi, nrm, ptrs6(tau_), derr_, stream_);
    time_end(0);
    if (!mq_zero) {
      launch_sum6(g_, ptrs6(tau_), false, partial_, dscal_ + kSlotMean, stream_);
      FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean, dscal_ + kSlotMean, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipStreamSynchronize(stream_));
      for (int c = 0; c < 6; ++c) F0[c] = hscal_[kSlotMean + c] / (double)nglobal_;
    }
    time_begin(1);
    launch_div(g_, ptrs6(tau_), ptrs3(fu_), XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
    time_end(1);
  }
  fft_g0_chain(fu_, -1.0, nullptr, tau_);   // the polarisation is dead once its divergence is taken

  // applyBCProjector  F:20247-20270: R = alpha*(bc_relax*MQ:F0 - (1-bc_relax)*M:(QC0:F00))
  Vec6 E, R;
  double t1[6], t2[6], t3[6];
  voigt_mv(BC_MQ_, F0, t1);
  voigt_mv(BC_QC0_, F00_, t2);
  voigt_mv(BC_M_, t2, t3);
  bool add_R = false;
  for (int c = 0; c < 6; ++c) {
    E.v[c] = E6[c];
    R.v[c] = mq_zero && opt_.bc_relax == 1.0 ? 0.0 : alpha * (opt_.bc_relax * t1[c] - (1 - opt_.bc_relax) * t3[c]);
    if (R.v[c] != 0.0) add_R = true;
  }
  time_begin(9);
  launch_eps_norm(g_, ptrs3(fu_), ptrs6(dst), E, R, add_R, partial_, dscal_ + kSlotSumSq,
                  XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
  time_end(9);
  if (timing_) times_.count++;
  // fu_ now holds the displacement this strain was built from (eps = E + sym grad u when R == 0)
  u_valid_ = !add_R && dst == eps_;
  if (dst == eps_) eps_stale_ = false;
  for (int c = 0; c < 6; ++c) E_cur_[c] = E6[c];
}

// z and y transforms of a plane in one kernel (grids whose complex z-y plane fits the LDS): option plane_fft (-1 = where
// available, 0 = off for A/B runs)
bool Solver::plane_fft_on() const {
  return opt_.plane_fft != 0 && nranks_ == 1 && fft_->can_plane();
}

// One chunked pair of the transform chain: first(c) then second(c) for runs c of `pc` x planes.
template <class A, class B>
void Solver::run_pairs(int pc, A first, B second) {
  for (int x0 = 0; x0 < g_.nx; x0 += pc) {
    const int np = std::min(pc, g_.nx - x0);
    first(x0, np);
    second(x0, np);
  }
}

void Solver::fft_g0_chain(double* buf, double alpha, const double* c12, double* xscratch) {  // alpha = -1: GammaOperator(..., -1)  F:20575
  fft_->set_joint_x(opt_.joint_x != 0);
  if (opt_.mode == 1) {
    // G0OperatorStaggeredHeat  F:20118-20135 on one component: fftVector(., 1), c1 = c10/|k|^2, fftInvVector
    const double scale = 1 / (double)nglobal_;
    const bool has_x = g_.nx > 1, has_y = g_.ny > 1;
    const bool plane = plane_fft_on() && has_x;
    const int pc1 = pair_chunk_planes(1);
    if (plane) {
      time_begin(2);
      fft_->zy_plane(buf, 1, g_.n, -1);
      time_end(2);
    } else if (pc1) {
      time_begin(2);
      run_pairs(pc1, [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->r2c_z(buf, 1, g_.n, &w); },
                [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2c_y(buf, 1, g_.n, -1, 1.0, &w); });
      time_end(2);
    } else {
      time_begin(2);
      fft_->r2c_z(buf, 1, g_.n);
      time_end(2);
      time_begin(3);
      fft_->c2c_y(buf, 1, g_.n, -1, (has_x || !has_y) ? 1.0 : scale);
      time_end(3);
    }
    const double c10 = -alpha / (2 * opt_.mu_0);  // G0OperatorFourierStaggeredHeat  F:19759-19764
    if (opt_.fuse_x && fft_->can_fuse(0, 1) && has_x) {
      // x transform, 1/N, scalar Green operator and inverse x transform in one kernel
      G0Params gp;
      for (int a = 0; a < 3; ++a) gp.kpm[a] = g0_kpm_[a], gp.kp[a] = g0_kp_[a];
      gp.c10 = c10;
      gp.c20 = 0.0;
      gp.inv_h0 = 2.0 * nxg_ / g_.dx;
      time_begin(5);
      fft_->fused_g0(buf, g_.n, 0, scale, gp, 0, 1);
      time_end(5);
    } else {
      time_begin(4);
      fft_->c2c_x(buf, 1, g_.n, -1, has_x ? scale : 1.0);
      time_end(4);
      if (!has_x && !has_y) fft_->scale(buf, 1, g_.n, scale);
      G0Tables tb;
      for (int a = 0; a < 3; ++a) {
        tb.kpm[a] = g0_kpm_[a];
        tb.kp[a] = g0_kp_[a];
      }
      time_begin(5);
      launch_g0_heat(g_, buf, tb, c10, stream_);
      time_end(5);
      time_begin(6);
      fft_->c2c_x(buf, 1, g_.n, +1, 1.0);
      time_end(6);
    }
    if (plane) {
      time_begin(8);
      fft_->zy_plane(buf, 1, g_.n, +1);
      time_end(8);
      return;
    }
    if (pc1) {
      time_begin(7);
      run_pairs(pc1, [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2c_y(buf, 1, g_.n, +1, 1.0, &w); },
                [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2r_z(buf, 1, g_.n, &w); });
      time_end(7);
      return;
    }
    time_begin(7);
    fft_->c2c_y(buf, 1, g_.n, +1, 1.0);
    time_end(7);
    time_begin(8);
    fft_->c2r_z(buf, 1, g_.n);
    time_end(8);
    return;
  }
  bool fuse_x = false, plane = false;
  // x-contiguous layout for the fused pass: on by size (fields beyond the Infinity Cache, where the fused pass's tile of nx
  // segments 2 MB apart is what bounds it) unless the option x_layout says otherwise
  const int xl_opt = opt_.x_layout;
  const bool xl = xscratch && opt_.fuse_x && fft_->can_fuse(0) && fft_->can_xlayout() && g_.nx > 1 && g_.ny > 1 &&
                  (xl_opt > 0 || (xl_opt < 0 && 3.0 * (double)g_.n * sizeof(double) > 1024.0 * 1024 * 1024));
  const int pc = pair_chunk_planes(3);
  if (xl) {
    if (pc) {   // r2c(c) -> y(c) per run of x planes: the spectrum of a chunk is still in the Infinity Cache when the y pass reads it
      time_begin(2);
      run_pairs(pc, [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->r2c_z(buf, 3, g_.n, &w); },
                [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2c_y_xlayout(buf, g_.n, xscratch, g_.n, 3, -1, 1.0, &w); });
      time_end(2);
    } else {
      time_begin(2);
      fft_->r2c_z(buf, 3, g_.n);
      time_end(2);
      time_begin(3);
      fft_->c2c_y_xlayout(buf, g_.n, xscratch, g_.n, 3, -1, 1.0);
      time_end(3);
    }
    G0Params gp;
    gp.c10 = -alpha / (opt_.mu_0);
    gp.c20 = -alpha / (opt_.mu_0 * (1 + opt_.mu_0 / (opt_.lambda_0 + opt_.mu_0)));
    gp.inv_h0 = 2.0 * nxg_ / g_.dx;
    if (c12) gp.c10 = c12[0], gp.c20 = c12[1];
    for (int a = 0; a < 3; ++a) gp.kpm[a] = g0_kpm_[a], gp.kp[a] = g0_kp_[a];
    time_begin(5);
    fft_->fused_g0(xscratch, g_.n, 0, 1 / (double)nglobal_, gp, 0, 3, 31, 0, true);
    time_end(5);
    if (pc) {
      time_begin(7);
      run_pairs(pc, [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2c_y_xlayout(xscratch, g_.n, buf, g_.n, 3, +1, 1.0, &w); },
                [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2r_z(buf, 3, g_.n, &w); });
      time_end(7);
      return;
    }
    time_begin(7);
    fft_->c2c_y_xlayout(xscratch, g_.n, buf, g_.n, 3, +1, 1.0);
    time_end(7);
    time_begin(8);
    fft_->c2r_z(buf, 3, g_.n);
    time_end(8);
    return;
  }
  {
    // fftVector  F:18481-18510: r2c in z, c2c in y, c2c in x; the 1/N of F:18501-18506 rides on the last pass
    const double scale = 1 / (double)nglobal_;
    const bool has_x = g_.nx > 1, has_y = g_.ny > 1;
    plane = plane_fft_on() && has_x;
    if (plane) {   // small grids: z and y transforms of a plane in one kernel
      time_begin(2);
      fft_->zy_plane(buf, 3, g_.n, -1);
      time_end(2);
    } else if (pc) {
      time_begin(2);
      run_pairs(pc, [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->r2c_z(buf, 3, g_.n, &w); },
                [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2c_y(buf, 3, g_.n, -1, 1.0, &w); });
      time_end(2);
    } else {
      time_begin(2);
      fft_->r2c_z(buf, 3, g_.n);
      time_end(2);
      time_begin(3);
      fft_->c2c_y(buf, 3, g_.n, -1, (has_x || !has_y) ? 1.0 : scale);
      time_end(3);
    }
    fuse_x = opt_.fuse_x && fft_->can_fuse(0);
    if (!fuse_x) {
      time_begin(4);
      fft_->c2c_x(buf, 3, g_.n, -1, has_x ? scale : 1.0);
      time_end(4);
      if (!has_x && !has_y) fft_->scale(buf, 3, g_.n, scale);
    }
  }
  {
    // G0OperatorFourierStaggered  F:19749-19755
    G0Params gp;
    gp.c10 = -alpha / (opt_.mu_0);
    gp.c20 = -alpha / (opt_.mu_0 * (1 + opt_.mu_0 / (opt_.lambda_0 + opt_.mu_0)));
    gp.inv_h0 = 2.0 * nxg_ / g_.dx;
    if (c12) gp.c10 = c12[0], gp.c20 = c12[1];
    G0Tables tb;
    for (int a = 0; a < 3; ++a) {
      tb.kpm[a] = gp.kpm[a] = g0_kpm_[a];
      tb.kp[a] = gp.kp[a] = g0_kp_[a];
    }
    time_begin(5);
    if (fuse_x) {
      // x transform, 1/N, Green operator and inverse x transform in one kernel (spectrum stays in registers)
      fft_->fused_g0(buf, g_.n, 0, 1 / (double)nglobal_, gp, 0);
    } else {
      launch_g0(g_, ptrs3(buf), tb, gp.c10, gp.c20, G0Layout{0, 0, 0}, stream_);
    }
    time_end(5);
  }
  if (!fuse_x) {
    time_begin(6);
    fft_->c2c_x(buf, 3, g_.n, +1, 1.0);
    time_end(6);
  }
  if (plane) {
    time_begin(8);
    fft_->zy_plane(buf, 3, g_.n, +1);
    time_end(8);
    return;
  }
  if (pc) {
    time_begin(7);
    run_pairs(pc, [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2c_y(buf, 3, g_.n, +1, 1.0, &w); },
              [&](int x0, int np) { const PlaneWindow w = {x0, np}; fft_->c2r_z(buf, 3, g_.n, &w); });
    time_end(7);
    return;
  }
  time_begin(7);
  fft_->c2c_y(buf, 3, g_.n, +1, 1.0);
  time_end(7);
  time_begin(8);
  fft_->c2r_z(buf, 3, g_.n);
  time_end(8);
}

// Planes per chunk of the paired z / y passes (0 = whole-field passes, the default).  Measured in round 6 and NOT faster:
// a bytes-only pair of copy passes gains 1.32 x from 128-MB chunks (tools/mall_chunk_probe.hip), the transform passes lose
// 3-10 % at every chunk size (EXPERIMENTS.md, round 6) -- they are bound by the requests a workgroup keeps in flight, which an
// Infinity-Cache hit does not shorten.  Kept as an option: same kernels on the same lines, bit-identical iterates.
int Solver::pair_chunk_planes(int) const {
  if (opt_.pair_chunk <= 0 || !fft_->can_window() || g_.nx < 2 || g_.ny < 2 || plane_fft_on()) return 0;
  return std::min(opt_.pair_chunk, g_.nx);
}

// ------------------------------------------------------------------ displacement-based pass
// eps_k = E + sym grad u_k is implied by the displacement of the previous pass, so the loop can carry
// u (3 components) instead of eps (6): one sweep turns u_k into the sums of squares of eps_k (the
// error estimator of pass k) and f_{k+1} = div (C - C0):eps_k, then the transform chain gives u_{k+1}.
// Same values as strain operator + polarisation + divergence (bit for bit), 2 kernels fewer per pass.
ScalarParams Solver::scalar_params(double mu_0, double alpha) const {
  ScalarParams sp;
  sp.n = pt_.n;
  for (int q = 0; q < kMaxPhases; ++q) sp.mu[q] = q < pt_.n ? pt_.mu[q] : 0.0;
  sp.alpha = alpha;
  sp.beta = -alpha * 2 * mu_0;  // calcStress  F:18138
  return sp;
}

FieldPtrs<kMaxPhases> Solver::phase_ptrs() const {
  FieldPtrs<kMaxPhases> phi;
  for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
  return phi;
}

bool Solver::u_loop_eligible(bool allow_mixed_bc) const {
  if (opt_.mode == 1) {
    // the scalar modes only have the potential-based loop
    if (nranks_ != 1) throw std::runtime_error("heat / porous mode is not available on slab-decomposed solvers");
    if (opt_.mixing != kMixVoigt) throw std::runtime_error("heat / porous mode supports Voigt mixing only");
    if (opt_.bc_relax != 1.0) throw std::runtime_error("heat / porous mode does not support bc_relax != 1");
    if (frobenius(BC_MQ_) >= kEps && !allow_mixed_bc)
      throw std::runtime_error("heat / porous mode: mixed boundary conditions run with method=basic (fg_run_load_case) only");
    return pt_.n >= 1;
  }
  if (!(opt_.u_loop && opt_.mode == 0 && opt_.gamma_scheme == 0 && nranks_ == 1 && pt_.n >= 1 &&
        (opt_.mixing == kMixVoigt || (opt_.mixing == kMixLaminate && normals_)) && opt_.bc_relax == 1.0))
    return false;
  if (frobenius(BC_MQ_) < kEps) return true;
  // mixed boundary conditions: <tau> of every pass corrects the prescribed mean of the next one; the tiled Voigt sweep
  // delivers it with the norms (run() only: the correction needs the host between passes)
  return allow_mixed_bc && opt_.u_loop >= 2 && opt_.u_tile && u_tile_supported(g_);   // Voigt, or laminate as its correction
}

// Two phases whose fractions are complementary bit for bit (phi_0 == 1 - phi_1: what normalizePhi leaves for two
// materials), checked once per geometry on the device.  Option phi_sweep = 0 switches the shortcut off (A/B runs).
bool Solver::two_phase_complementary() {
  if (pt_.n != 2 || opt_.mode != 0 || !opt_.phi_sweep) return false;
  if (complement_dirty_) {
    int one = 1;
    FG_HIP_CHECK(hipMemcpyAsync(derr_ + 1, &one, sizeof(int), hipMemcpyHostToDevice, stream_));
    launch_complement_check(g_, phi_, phi_ + g_.n, derr_ + 1, stream_);
    int flag = 0;
    FG_HIP_CHECK(hipMemcpyAsync(&flag, derr_ + 1, sizeof(int), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    complementary_ = flag != 0;
    complement_dirty_ = false;
  }
  return complementary_;
}

// A = sum_p phi_p 2 mu_p, B = sum_p phi_p lambda_p per voxel (k_effective_moduli), computed once per geometry
FieldPtrs<2> Solver::effective_moduli() {
  if (!mod_) {
    FG_HIP_CHECK(hipMalloc(&mod_, 2 * (size_t)g_.n * sizeof(double)));
    mod_dirty_ = true;
  }
  FieldPtrs<2> mod;
  mod.p[0] = mod_;
  mod.p[1] = mod_ + g_.n;
  if (mod_dirty_) {
    PhaseTable t = phase_table();
    if (opt_.mode == 1)   // scalar modes: k_effective_moduli stores sum phi 2 mu, the sweep wants sum phi mu
      for (int q = 0; q < kMaxPhases; ++q) t.mu[q] = 0.5 * pt_.mu[q], t.lambda[q] = 0.0;
    launch_effective_moduli(g_, t, phase_ptrs(), mod, stream_);
    mod_dirty_ = false;
  }
  return mod;
}

void Solver::build_laminate_lists() {
  if (!mixed_dirty_) return;
  FieldPtrs<kMaxPhases> phi;
  for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
  for (void* p : {(void*)mixed_list_, (void*)aff_list_, (void*)aff_slots_, (void*)dtau_})
    if (p) FG_HIP_CHECK(hipFree(p));
  mixed_list_ = aff_list_ = nullptr;
  aff_slots_ = nullptr;
  dtau_ = nullptr;
  aff_n_ = 0;
  for (void* p : {(void*)lam_phic_, (void*)lam_nrmc_, (void*)lam_epsc_})
    if (p) FG_HIP_CHECK(hipFree(p));
  lam_phic_ = lam_nrmc_ = lam_epsc_ = nullptr;
  mixed_n_ = launch_mixed_list(g_, pt_.n, phi, &mixed_list_, stream_);
  if (mixed_n_) {
    FG_HIP_CHECK(hipMalloc(&dtau_, (size_t)mixed_n_ * 6 * sizeof(double)));
    // compact static copies for the interface solve (k_interface_solve) and its strain scratch
    const size_t n = mixed_n_;
    FG_HIP_CHECK(hipMalloc(&lam_phic_, n * pt_.n * sizeof(double)));
    FG_HIP_CHECK(hipMalloc(&lam_nrmc_, n * 3 * sizeof(double)));
    FG_HIP_CHECK(hipMalloc(&lam_epsc_, n * 6 * sizeof(double)));
    FieldPtrs<3> nrm;
    for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
    launch_interface_static(g_, pt_.n, phi, nrm, mixed_list_, mixed_n_, lam_phic_, lam_nrmc_, stream_);
    // x-slab: no x neighbours across the slab faces in the slots (gu_: the grid with the slab's plane mapping)
    aff_n_ = launch_affected_list(slab_layout_ ? gu_ : g_, mixed_list_, mixed_n_, &aff_list_, &aff_slots_, stream_);
  }
  mixed_dirty_ = false;
}

void Solver::u_pass_front(const double* E6) {
  FieldPtrs<kMaxPhases> phi;
  for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
  // the strain that belongs to u_k was built with the prescribed strain of ITS pass (E_cur_);
  // E6 is the prescribed strain of the pass being started and takes effect with u_{k+1}
  Vec6 E;
  for (int c = 0; c < 6; ++c) {
    E.v[c] = E_cur_[c];
    E_next_[c] = E6[c];
  }
  time_begin(0);
  if (opt_.mode == 1) {
    if (opt_.u_loop >= 2) {
      // fast variant: effective conductivity a = sum_p phi_p mu_p precomputed (first moduli array)
      const bool mixed = !(frobenius(BC_MQ_) < kEps) && in_run_;
      sc_tau_sums_ = launch_sc_sweep_fast(g_, opt_.mu_0, fu_, effective_moduli().p[0], fu_alt_, E, partial_, dscal_ + kSlotSumSq,
                                          stream_, mixed ? dscal_ + kSlotMean : nullptr);
    } else {
      sc_tau_sums_ = false;
      launch_sc_sweep(g_, scalar_params(opt_.mu_0, 1.0), fu_, phase_ptrs(), fu_alt_, E, partial_, dscal_ + kSlotSumSq, stream_);
    }
  } else if (opt_.mixing != kMixVoigt && opt_.u_loop < 2) {
    // laminate mixing, exact order: strain + polarisation from u in one sweep (tau stored), divergence as its own sweep
    FieldPtrs<3> nrm;
    for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
    launch_u_stress(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs3(fu_), phi, nrm, ptrs6(tau_), E, partial_,
                    dscal_ + kSlotSumSq, derr_, stream_);
    time_end(0);
    time_begin(1);
    launch_div(g_, ptrs6(tau_), ptrs3(fu_alt_), XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
    time_end(1);
    eps_stale_ = true;
    return;
  } else if (opt_.u_loop >= 2) {
    // fast variant: per-voxel effective moduli instead of the per-phase accumulation (computed once per geometry, on
    // first use: the tiled sweep of two complementary phases does without them)
    const bool laminate = opt_.mixing != kMixVoigt;
    if (laminate) {
      build_laminate_lists();
      if (opt_.laminate_overlap) {   // fork: u_k is complete, the previous pass is done with d
        if (!aux_stream_) {
          int lo = 0, hi = 0;
          FG_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = numerically lowest = highest priority
          FG_HIP_CHECK(hipStreamCreateWithPriority(&aux_stream_, hipStreamNonBlocking, hi));
          FG_HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
          FG_HIP_CHECK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
        }
        FG_HIP_CHECK(hipEventRecord(ev_fork_, stream_));
      }
    }
    if (opt_.u_tile && u_tile_supported(g_)) {
      const bool sum_tau = !(frobenius(BC_MQ_) < kEps);   // mixed BC: sums of tau land in kSlotMean
      if (two_phase_complementary()) {
        // two phases with phi_0 = 1 - phi_1: the sweep reads phi_1 and forms the moduli itself (8 B per voxel less)
        FieldPtrs<2> ph;
        ph.p[0] = phi_ + g_.n;
        ph.p[1] = nullptr;
        const PhaseTable t = phase_table();
        launch_u_tile(g_, opt_.mu_0, opt_.lambda_0, ptrs3(fu_), ph, ptrs3(fu_alt_), E, partial_, dscal_ + kSlotSumSq, stream_, sum_tau,
                      &t);
      } else {
        launch_u_tile(g_, opt_.mu_0, opt_.lambda_0, ptrs3(fu_), effective_moduli(), ptrs3(fu_alt_), E, partial_,
                      dscal_ + kSlotSumSq, stream_, sum_tau);
      }
    } else   // grids the tiles do not fit (odd nz, short rows): the untiled sweep
      launch_u_fast(g_, opt_.mu_0, opt_.lambda_0, ptrs3(fu_), effective_moduli(), ptrs3(fu_alt_), E, partial_, dscal_ + kSlotSumSq,
                    stream_);
    if (laminate) {
      // laminate mixing = the Voigt sweep over all voxels + the divergence of (tau_laminate - tau_voigt), which lives
      // on the interface voxels (lists built once per geometry, see k_interface_strain)
      FieldPtrs<3> nrm;
      for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
      // d depends on u only: its two kernels (a light gather, a short solve) run on a second stream beside the sweep
      hipStream_t ds = opt_.laminate_overlap ? aux_stream_ : stream_;
      if (ds != stream_) FG_HIP_CHECK(hipStreamWaitEvent(ds, ev_fork_, 0));
      launch_interface_delta(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs3(fu_), E, mixed_list_, mixed_n_, lam_epsc_,
                             lam_phic_, lam_nrmc_, dtau_, derr_, ds);
      if (ds != stream_) {
        FG_HIP_CHECK(hipEventRecord(ev_join_, ds));
        FG_HIP_CHECK(hipStreamWaitEvent(stream_, ev_join_, 0));
      }
      launch_delta_div(g_, aff_list_, aff_slots_, aff_n_, dtau_, ptrs3(fu_alt_), stream_);
      if (!(frobenius(BC_MQ_) < kEps))   // mixed BC: <tau_laminate> = <tau_voigt> (from the sweep) + sum of the differences / N
        launch_sum_dtau(dtau_, mixed_n_, partial_, dscal_ + kSlotScratch, stream_);
    }
  } else {
    launch_u_stress_div_voigt(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs3(fu_), phi, ptrs3(fu_alt_), E,
                              partial_, dscal_ + kSlotSumSq, stream_);
  }
  time_end(0);
  eps_stale_ = true;
}

void Solver::u_pass_back() {
  back_ready_ = false;
  launch_pending_back();
  adopt_back();
}

void Solver::ensure_eps() {
  if (slab_layout_ && su_valid_ && eps_stale_) {   // slab driver: the state is su_[cur] with its halo planes
    slab_materialise_eps();
    return;
  }
  if (!eps_stale_ || !u_valid_) return;
  Vec6 E, R;
  for (int c = 0; c < 6; ++c) E.v[c] = E_cur_[c], R.v[c] = 0.0;
  if (opt_.mode == 1) {
    launch_sc_grad(g_, fu_, ptrs3(eps_), E, partial_, dscal_ + kSlotScratch, stream_);
    eps_stale_ = false;
    return;
  }
  launch_eps_norm(g_, ptrs3(fu_), ptrs6(eps_), E, R, false, partial_, dscal_ + kSlotScratch,
                  XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
  eps_stale_ = false;
}

void Solver::iterate(const double* E6, int n) {
  FG_HIP_CHECK(hipSetDevice(device_));
  int i = 0;
  if (u_loop_eligible()) {
    if (!u_valid_ && n > 0 && opt_.mode == 1) {
      // no potential yet: the zero gradient field is E = 0, T = 0
      FG_HIP_CHECK(hipMemsetAsync(fu_, 0, g_.n * sizeof(double), stream_));
      for (int c = 0; c < 6; ++c) E_cur_[c] = 0.0;
      u_valid_ = true;
      eps_stale_ = true;
    }
    if (!u_valid_ && n > 0) {
      ensure_eps();
      basic_scheme(E6);  // leaves u in fu_ and eps in eps_
      ++i;
    }
    for (; i < n; ++i) {
      u_pass_front(E6);
      u_pass_back();
    }
    return;
  }
  for (; i < n; ++i) basic_scheme(E6);
}

// ------------------------------------------------------------------ means
void Solver::mean_stress(double* out6) {
  FG_HIP_CHECK(hipSetDevice(device_));
  ensure_eps();
  if (pt_.n < 1) throw std::runtime_error("No materials specified");
  if (opt_.mode == 1) {
    launch_sc_flux_mean(g_, scalar_params(0.0, 1.0 / (double)nglobal_), ptrs3(eps_), phase_ptrs(), partial_,
                        dscal_ + kSlotMean, stream_);
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean, dscal_ + kSlotMean, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    for (int c = 0; c < 6; ++c) out6[c] = c < 3 ? hscal_[kSlotMean + c] : 0.0;
    return;
  }
  FieldPtrs<kMaxPhases> phi;
  for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
  FieldPtrs<3> nrm;
  for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
  // meanPK1: alpha /= nxyz, accumulate  F:12318-12340
  launch_stress_mean(g_, stress_params(0.0, 0.0, 1.0 / (double)nglobal_), ptrs6(eps_), phi, nrm, partial_,
                     dscal_ + kSlotMean, derr_, stream_);
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean, dscal_ + kSlotMean, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  for (int c = 0; c < 6; ++c) out6[c] = hscal_[kSlotMean + c];
}

double Solver::mean_energy() {
  FG_HIP_CHECK(hipSetDevice(device_));
  ensure_eps();
  if (pt_.n < 1) throw std::runtime_error("No materials specified");
  if (opt_.mode == 1) throw std::runtime_error("the energy error estimator is not available in heat / porous mode");
  FieldPtrs<3> nrm;
  for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
  launch_energy_mean(g_, stress_params(0.0, 0.0, 1.0), ptrs6(eps_), phase_ptrs(), nrm, partial_, dscal_ + kSlotMean, derr_, stream_);
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean, dscal_ + kSlotMean, sizeof(double), hipMemcpyDeviceToHost, stream_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  return hscal_[kSlotMean] / (double)nglobal_;
}

// create_error_estimator  F:14940-14972 for the estimators that measure a mean of the strain field: their constructors run on
// the field the step starts from (zero for a first step: <sigma> = 0, <W> = 0), update() after every iteration
void Solver::estimator_begin(bool fresh) {
  double m[6] = {0, 0, 0, 0, 0, 0};
  if (opt_.error_estimator == 2) {
    if (!fresh) mean_stress(m);
    est_.start_sigma(m);
  } else if (opt_.error_estimator == 3) {
    est_.start_energy(fresh ? 0.0 : mean_energy());
  }
}

void Solver::estimator_update(double* abs_err, double* rel_err) {
  if (opt_.error_estimator == 2) {
    double m[6];
    mean_stress(m);
    est_.update_sigma(m, abs_err, rel_err);
  } else if (opt_.error_estimator == 3) {
    est_.update_energy(mean_energy(), abs_err, rel_err);
  } else if (opt_.error_estimator == 4) {
    *abs_err = *rel_err = 1.0;
  }
}

void Solver::mean_strain(double* out6) {
  FG_HIP_CHECK(hipSetDevice(device_));
  ensure_eps();
  launch_sum6(g_, ptrs6(eps_), false, partial_, dscal_ + kSlotMean, stream_);
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMean, dscal_ + kSlotMean, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  for (int c = 0; c < 6; ++c) out6[c] = hscal_[kSlotMean + c] / (double)nglobal_;
}

double Solver::volume_fraction(int p) {
  if (p < 0 || p >= pt_.n) throw std::runtime_error("phase index out of range");
  FG_HIP_CHECK(hipSetDevice(device_));
  launch_sum1(g_, phi_ + (long)p * g_.n, partial_, dscal_ + kSlotMisc, stream_);
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMisc, dscal_ + kSlotMisc, sizeof(double), hipMemcpyDeviceToHost, stream_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  return hscal_[kSlotMisc] / (double)nglobal_;
}

// calcRefMaterial  F:22283-22313 -> getRefMaterial  F:12153-12236
void Solver::calc_ref_material() {
  FG_HIP_CHECK(hipSetDevice(device_));
  if (pt_.n < 1) throw std::runtime_error("No materials specified");
  FieldPtrs<kMaxPhases> phi;
  for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
  if (opt_.mode == 1)
    launch_sc_minmax(g_, scalar_params(0.0, 1.0), phi, partial_, dscal_ + kSlotMinMax, stream_);
  else
    launch_tangent_minmax(g_, phase_table(), opt_.mixing, phi, partial_, dscal_ + kSlotMinMax, derr_, stream_);
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotMinMax, dscal_ + kSlotMinMax, 2 * sizeof(double), hipMemcpyDeviceToHost, stream_));
  check_device_error("reference material");
  double lambda_min = hscal_[kSlotMinMax], lambda_max = -hscal_[kSlotMinMax + 1];
  if (lambda_min < 0) lambda_min = 0;  // F:12183-12223
  double mu_0 = 0.5 * (lambda_min + lambda_max);
  mu_0 *= 0.5 * opt_.ref_scale;
  opt_.mu_0 = mu_0;
  recompute_bc();  // F:22312
}

// bc_error  F:21129-21161
double Solver::bc_error(const double* E_cur, const double* S_cur) {
  double Emean[6], Smean[6], PE[6], QS[6], PEc[6], d[6];
  mean_strain(Emean);
  mean_stress(Smean);
  voigt_mv(BC_P_, Emean, PE);
  voigt_mv(BC_Q_, Smean, QS);
  voigt_mv(BC_P_, E_cur, PEc);
  const double norm_E = voigt_norm2(PEc);
  for (int i = 0; i < 6; ++i) d[i] = PE[i] - E_cur[i];
  const double err_F = voigt_norm2(d) / ((norm_E < opt_.bc_tol) ? 1 : norm_E);
  const double norm_S = voigt_norm2(S_cur);
  for (int i = 0; i < 6; ++i) d[i] = QS[i] - S_cur[i];
  const double err_S = voigt_norm2(d) / ((norm_S < opt_.bc_tol) ? 1 : norm_S);
  return err_F > err_S ? err_F : err_S;
}

// ------------------------------------------------------------------ the solver loop
// run F:21247-21398 -> runLoadsteppingSolver (single load step, t = 1) F:21584-21685
// -> runSolver -> runBasic F:21716-21805 with the stop rule of _converged F:21177-21244.
bool Solver::run(const double* E6, const double* S6) {
  const double one = 1.0;
  return run_load_steps(E6, S6, &one, 1, 0, nullptr, nullptr);
}

// norm of the current strain field as EpsilonErrorEstimator's constructor takes it (F:14612-14618): a load step that
// continues from the previous one starts its error estimate there, the first one at the zero field
double Solver::current_norm9() {
  ensure_eps();
  launch_sum6(g_, ptrs6(eps_), true, partial_, dscal_ + kSlotScratch, stream_);
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotScratch, dscal_ + kSlotScratch, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  double s9 = 0.0;
  const int nc = opt_.mode == 1 ? 3 : 6;
  for (int c = 0; c < nc; ++c) {
    const double m = std::sqrt(hscal_[kSlotScratch + c] / (double)nglobal_);
    s9 += m * m * ((c >= 3) ? 2.0 : 1.0);   // shear norms mirrored to 9 entries (fix_dim)
  }
  return std::sqrt(s9);
}

// runLoadsteppingSolver  F:21584-21685: the prescribed values are scaled by the parameter of every load step, every
// step continues from the strain field of the one before (the field is zeroed once, in run(), F:21379), and after each
// step the load-step action runs (performLoadstepActions F:21435-21447: the caller's callback, non-zero = stop).
// Load-step extrapolation (loadstep_extrapolation_order > 0, polynomial method F:21468-21514): from the second step on the
// step starts from the polynomial through the converged strain fields of the last order + 1 steps evaluated at its parameter.
bool Solver::run_load_steps(const double* E6, const double* S6, const double* params, int nparams, int first,
                            LoadstepCallback step_cb, void* user) {
  FG_HIP_CHECK(hipSetDevice(device_));
  if (pt_.n < 1) throw std::runtime_error("No materials specified");
  if (nparams < 1 || first < 0 || !params) throw std::runtime_error("invalid load steps");
  solve_time_ = 0.0;
  cg_u_active_ = false;
  residuals_.clear();
  iterations_ = 0;
  cancel_ = false;
  double Emax[6], Smax[6];
  for (int i = 0; i < 6; ++i) {
    Emax[i] = E6[i];
    Smax[i] = S6 ? S6[i] : 0.0;
  }
  recompute_bc();  // F:21354
  {
    const double se = std::sqrt(kEps);
    double t[6];
    voigt_mv(BC_P_, Smax, t);
    if (norm2(t, 6) > se * norm2(Smax, 6)) throw std::runtime_error("Incompatible stress boundary condition specified");
    voigt_mv(BC_Q_, Emax, t);
    if (norm2(t, 6) > se * norm2(Emax, 6)) throw std::runtime_error("Incompatible strain boundary condition specified");
  }
  FG_HIP_CHECK(hipMemsetAsync(eps_, 0, 6 * (size_t)g_.n * sizeof(double), stream_));  // F:21379
  u_valid_ = false;
  eps_stale_ = false;
  // converged strain fields of the last steps (device copies, oldest first)  F:21586
  struct Kept {
    double t;
    double* eps;
  };
  std::vector<Kept> last;
  struct Release {
    std::vector<Kept>& v;
    ~Release() {
      for (Kept& k : v) (void)hipFree(k.eps);
    }
  } release{last};
  const size_t f6 = 6 * (size_t)g_.n * sizeof(double);
  for (int istep = first; istep < nparams; ++istep) {
    double E[6], S[6];
    for (int i = 0; i < 6; ++i) E[i] = params[istep] * Emax[i], S[i] = params[istep] * Smax[i];
    fresh_step_ = istep == first;
    const int order = opt_.loadstep_extrapolation_order;
    if (order > 0 && istep > first) {   // F:21634-21650
      if (slab_layout_) throw std::runtime_error("load-step extrapolation is not available on slab-decomposed solvers");
      double* buf = nullptr;
      while ((int)last.size() > order) {   // the oldest buffer is reused for the new copy
        if (buf) FG_HIP_CHECK(hipFree(buf));
        buf = last.front().eps;
        last.erase(last.begin());
      }
      if (!buf) FG_HIP_CHECK(hipMalloc(&buf, f6));
      ensure_eps();
      FG_HIP_CHECK(hipMemcpyAsync(buf, eps_, f6, hipMemcpyDeviceToDevice, stream_));
      last.push_back(Kept{params[istep - 1], buf});
      const int n = (int)last.size();
      if (n >= 2) {
        // w = V^-T tpowers with V_ij = t_i^j: solve V^T w = tpowers (Gauss elimination with partial pivoting, n <= 8)
        double A[8][9], w[8];
        for (int i = 0; i < n; ++i) {
          for (int j = 0; j < n; ++j) A[i][j] = std::pow(last[j].t, i);   // (V^T)_ij = t_j^i
          A[i][n] = std::pow(params[istep], i);
        }
        for (int c = 0; c < n; ++c) {
          int piv = c;
          for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
          if (A[piv][c] == 0.0) throw std::runtime_error("Error inverting Vandermonde matrix");   // F:21490-21492
          for (int j = 0; j <= n; ++j) std::swap(A[c][j], A[piv][j]);
          for (int r = c + 1; r < n; ++r) {
            const double m = A[r][c] / A[c][c];
            for (int j = c; j <= n; ++j) A[r][j] -= m * A[c][j];
          }
        }
        for (int r = n - 1; r >= 0; --r) {
          double v = A[r][n];
          for (int j = r + 1; j < n; ++j) v -= A[r][j] * w[j];
          w[r] = v / A[r][r];
        }
        const double* in[8];
        for (int i = 0; i < n; ++i) in[i] = last[i].eps;
        launch_lincomb(n, in, w, eps_, 6 * g_.n, stream_);
        u_valid_ = false;      // the state is the extrapolated strain field: one strain-state pass, then the displacement loop
        eps_stale_ = false;
      }
    }
    if (run_one_step(E, S)) return true;
    if (step_cb && step_cb(user, istep)) return true;   // "Loadstep callback break request."
  }
  return false;
}

// runSolver  F:21400-21433 for one load step
bool Solver::run_one_step(const double* E0, const double* S0) {
  // EpsilonErrorEstimator  F:14591-14637: constructed on the field the step starts from (zero for the first step)
  const double prev0 = fresh_step_ ? 0.0 : current_norm9();
  if (opt_.error_estimator >= 2) {
    if (opt_.mode == 1) throw std::runtime_error("heat / porous mode supports the error estimators epsilon and residual");
    estimator_begin(fresh_step_);
  }
  if (opt_.method == 1 && opt_.mode == 1) {
    (void)u_loop_eligible();   // throws for configurations the scalar modes do not support
    return run_cg_scalar(E0, prev0);
  }
  if (opt_.method == 1) return run_cg(E0, S0, prev0);
  if (opt_.error_estimator == 1)   // ErrorEstimator::update  F:14359
    throw std::runtime_error("Selected error estimator is not compatible with the selected solution method");
  const double t_start = now_seconds();
  for (int i = 0; i < 6; ++i) F00_[i] = 0.0;
  // Displacement-based loop: eps_0 = 0 gives tau = 0, u_1 = 0 and eps_1 = E, so the loop starts from
  // u = 0 and every pass is [u_k -> norms of eps_k, f_{k+1}] + [f_{k+1} -> u_{k+1}] (see u_pass_front).
  bool uloop = u_loop_eligible(true);
  const bool mixed_bc = !(frobenius(BC_MQ_) < kEps);   // (NaN until calcRefMaterial ran: Q != 0)
  if (uloop && fresh_step_) {
    FG_HIP_CHECK(hipMemsetAsync(fu_, 0, 3 * g_.n * sizeof(double), stream_));
    u_valid_ = true;
    eps_stale_ = true;
    for (int i = 0; i < 6; ++i) E_cur_[i] = E0[i];
  }
  in_run_ = true;

  double prev = prev0;
  long iter = 1;
  // a continuing load step in the displacement loop: the state is u of the previous step (eps = E_old + sym grad u); one
  // unrecorded pass turns it into u' with eps' = E_new + sym grad u' -- the field the reference's first iteration of the
  // step produces -- and the loop below then measures eps' in its first pass
  bool carry = uloop && !fresh_step_ && u_valid_;
  bool update_ref = opt_.update_ref != 0;
  double E[6];
  for (int i = 0; i < 6; ++i) E[i] = E0[i];
  bool failed = false;
  const double small = std::numeric_limits<double>::min();

  for (;;) {
    if (update_ref) {
      calc_ref_material();
      // calcBCMean  F:20242-20245
      double t1[6], t2[6], t3[6];
      voigt_mv(BC_QC0_, E0, t1);
      for (int i = 0; i < 6; ++i) t2[i] = S0[i] - t1[i];
      voigt_mv(BC_M_, t2, t3);
      for (int i = 0; i < 6; ++i) E[i] = E0[i] + opt_.bc_relax * t3[i];
      update_ref = false;
    }
    bool pending_back = false;
    pending_back_ = back_ready_ = false;
    if (carry) {
      if (mixed_bc) {
        carry = false;   // (the correction of the prescribed mean needs <tau> on the host: take the strain-state pass)
        uloop = false;
      } else {
        u_pass_front(E);
        u_pass_back();
        carry = false;
      }
    }
    if (uloop && u_valid_) {
      if (iter == 1 && fresh_step_)
        for (int i = 0; i < 6; ++i) E_cur_[i] = E[i];  // eps_1 = E (u_1 = 0)
      u_pass_front(E);
      if (opt_.mode == 1 && mixed_bc && !sc_tau_sums_) {
        // heat / porous with mixed boundary conditions (initBCProjector in GammaOperatorStaggeredHeat F:20342-20350): the
        // LDS-tiled sweep leaves the sums of tau in kSlotMean; where it does not apply, <tau> = <P(g) - 2 mu0 g> is taken
        // from the gradient field (two extra sweeps per pass)
        ensure_eps();
        launch_sc_flux_mean(g_, scalar_params(opt_.mu_0, 1.0 / (double)nglobal_), ptrs3(eps_), phase_ptrs(), partial_,
                            dscal_ + kSlotMean, stream_);
        eps_stale_ = true;
      }
      pending_back = true;
      pending_back_ = true;   // fetch_norms_and_errors enqueues the FFT chain behind the copies
    } else {
      // no displacement behind the strain field (a step that starts from a given / extrapolated field, mixed boundary
      // conditions with their correction term in the strain): strain-state pass.  If it leaves a displacement, the loop
      // goes on in displacement space after the unrecorded pass of a continuing step (carry).
      basic_scheme(E);
      if (uloop && u_valid_ && !mixed_bc) carry = true;
      else uloop = false;
    }
    fetch_norms_and_errors("stress");
    if (pending_back && mixed_bc) {
      // applyBCProjector  F:20247-20270 with bc_relax = 1: eps_{k+1} = E + alpha MQ:<tau_k> + sym grad u_{k+1}
      double F0[6], t1[6];
      for (int c = 0; c < 6; ++c) F0[c] = hscal_[kSlotMean + c] / (double)nglobal_;
      if (opt_.mode == 1)   // components 3..5 do not exist; the flux-mean sweep has divided by N already, the tiled sweep has not
        for (int c = 0; c < 6; ++c) F0[c] = c < 3 ? hscal_[kSlotMean + c] / (sc_tau_sums_ ? (double)nglobal_ : 1.0) : 0.0;
      if (opt_.mixing != kMixVoigt)
        for (int c = 0; c < 6; ++c) F0[c] += hscal_[kSlotScratch + c] / (double)nglobal_;
      voigt_mv(BC_MQ_, F0, t1);
      for (int c = 0; c < 6; ++c) E_next_[c] = E[c] - t1[c];   // alpha = -1  (F:20575)
    }

    // component_norm + fix_dim + norm_2 over 9 mirrored entries  F:10127-10138, F:14600-14609, F:14627
    double m[6], s9 = 0.0;
    for (int c = 0; c < 6; ++c) {
      sumsq_[c] = hscal_[kSlotSumSq + c];
      m[c] = std::sqrt(sumsq_[c] / (double)nglobal_);
    }
    for (int c = 0; c < 6; ++c) s9 += m[c] * m[c];
    for (int c = 3; c < 6; ++c) s9 += m[c] * m[c];
    const double cur = std::sqrt(s9);
    double abs_err = std::fabs(prev - cur);
    double rel_err = abs_err / (small + cur);
    prev = cur;
    if (opt_.error_estimator >= 2) estimator_update(&abs_err, &rel_err);   // sigma / energy / none: F:14410-14587

    // _converged  F:21177-21244
    if (std::isnan(rel_err)) {
      failed = true;  // "NaN detected in solution. Aborting."
      break;
    }
    if (cancel_) {
      failed = true;
      break;
    }
    residuals_.push_back(rel_err);
    if (cb_ && cb_(cb_user_)) break;
    if (cancel_) {
      failed = true;
      break;
    }
    if (iter >= opt_.maxiter) break;
    if (rel_err <= opt_.tol || abs_err <= opt_.abs_tol) {
      const double bc_err = bc_error(E0, S0);
      if (bc_err <= opt_.bc_tol) break;
    }
    if (pending_back) adopt_back();
    iter++;
  }
  pending_back_ = back_ready_ = false;
  in_run_ = false;
  iterations_ = iter;
  ensure_eps();
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  solve_time_ += now_seconds() - t_start;
  return failed;
}

// ------------------------------------------------------------------ conjugate gradients
// runCGElasticity  F:23153-23247 on the operator  eps -> -Gamma0 : (C - C0) : eps  (krylovOperator
// F:20583-20587 = one basicScheme pass with E = 0), l2 inner product, epsilon error estimator.
// The same iteration carried in displacement space (see k_cgu_dot): eps = E + grad_s u_e, r / p / w = grad_s u_r / u_p / u_w.
//   u_e = fu_ (so the strain is materialised from it afterwards), u_w = fu_alt_ (output of the FFT chain),
//   u_r, u_p = the two halves of the 6-component buffer cg_r_.
// Per iteration: one displacement sweep (the K1 of the basic scheme with E = 0) + FFT chain, one gradient dot product,
// one point-wise update of u_e and u_r, one gradient norm sweep, one point-wise update of u_p.
bool Solver::run_cg_u(const double* E0, double prev0) {
  const double t_start = now_seconds();
  const size_t f3 = 3 * (size_t)g_.n * sizeof(double);
  if (!cg_r_) FG_HIP_CHECK(hipMalloc(&cg_r_, 2 * f3));
  double* u_r = cg_r_;
  double* u_p = cg_r_ + 3 * g_.n;
  // Fused form (option cg_fused, default where the tiled sweep fits; Voigt mixing): the vector work of an iteration is two
  // tiled sweeps instead of four kernels -- p:(p - w) (launch_cgu_tile mode 0), and the update of eps and r together with the
  // norms of the new eps and r:r (mode 1) -- and the direction update p = r + beta p is formed inside the operator's
  // displacement sweep (launch_u_tile_cg).  Updates are out of place (the tiles' halo rows re-evaluate them), so u_e, u_r and
  // u_p alternate between two buffers each; fu_ stays the current iterate.
  const int fused_opt = opt_.cg_fused;
  bool fused = fused_opt != 0 && opt_.u_tile && u_tile_supported(g_) && !slab_layout_;
  if (fused && (!cg_p_ || !fu_cg_)) {
    // nine more components (the alternates of u_e, u_r, u_p): on grids that fill the card (1024^3: 78 GB) the four-kernel form
    size_t free_b = 0, total_b = 0;
    FG_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (cg_p_ ? 0 : 2 * f3) + (fu_cg_ ? 0 : f3);
    if (free_b < need + (size_t)(0.02 * (double)total_b)) fused = false;
  }
  const bool fused_dir = fused && opt_.mixing == kMixVoigt;   // laminate mixing: the interface kernels read u_p as a stored field
  double *r_alt = nullptr, *p_alt = nullptr;
  if (fused) {
    if (!cg_p_) FG_HIP_CHECK(hipMalloc(&cg_p_, 2 * f3));
    if (!fu_cg_) FG_HIP_CHECK(hipMalloc(&fu_cg_, f3));
    r_alt = cg_p_;
    p_alt = cg_p_ + 3 * g_.n;
  }
  for (int i = 0; i < 6; ++i) F00_[i] = 0.0;
  in_run_ = true;
  cg_u_active_ = true;
  const double small = std::numeric_limits<double>::min();
  if (opt_.update_ref) calc_ref_material();
  Vec6 E, Z;
  for (int i = 0; i < 6; ++i) E.v[i] = E0[i], Z.v[i] = 0.0;   // Q = 0: calcBCMean leaves E0
  // the operator on a displacement: u -> f = div((C - C0)(Eadd + grad_s u)) -> FFT chain (alpha = -1) -> fu_alt_
  auto apply = [&](double* u_in, const double* Eadd) {
    double* keep = fu_;
    fu_ = u_in;
    for (int c = 0; c < 6; ++c) E_cur_[c] = Eadd[c];
    u_pass_front(Eadd);   // sweeps fu_ (with E_cur_) into fu_alt_; its norm sums are not used here
    fu_ = keep;
    fft_g0_chain(fu_alt_, -1.0, nullptr, tau_);
  };
  // fused form: u_p := u_r + beta u_p (beta from the sums at dscal_[i_num] / dscal_[i_den]) inside the sweep of the operator
  auto apply_dir = [&](int i_num, int i_den) {
    FieldPtrs<2> m;
    const PhaseTable t = phase_table();
    const bool two = two_phase_complementary();
    if (two) {
      m.p[0] = phi_ + g_.n;
      m.p[1] = nullptr;
    } else {
      m = effective_moduli();
    }
    time_begin(0);
    launch_u_tile_cg(g_, opt_.mu_0, opt_.lambda_0, ptrs3(u_p), ptrs3(u_r), ptrs3(p_alt), m, ptrs3(fu_alt_), Z, dscal_, i_num, i_den,
                     (double)nglobal_, std::numeric_limits<double>::min(), partial_, dscal_ + kSlotSumSq, stream_, two ? &t : nullptr);
    time_end(0);
    fft_g0_chain(fu_alt_, -1.0, nullptr, tau_);
    std::swap(u_p, p_alt);
  };
  int beta_num = 0, beta_den = 0;   // fused form: slots of the pending direction update
  // eps_0 = E (u_e = 0);  r = -Gamma0 (C - C0) E  (+ E - eps_0 = 0, adjustResidual F:10012-10022)
  // The CG scalars stay on the device (k_cgu_axpy forms alpha and beta from the sums the dot sweeps leave in
  // dscal_): the host fetches only the seven sums of the stop rule, and -- when no callback can look at the state
  // in between -- has already enqueued the next direction update and operator application when it waits for them.
  const int blk[2] = {kSlotCg, kSlotCg + 8}, s0 = kSlotCg + 16;
  const double nvox = (double)nglobal_;
  FG_HIP_CHECK(hipMemsetAsync(fu_, 0, f3, stream_));
  apply(fu_, E.v);
  FG_HIP_CHECK(hipMemcpyAsync(u_r, fu_alt_, f3, hipMemcpyDeviceToDevice, stream_));
  FG_HIP_CHECK(hipMemcpyAsync(u_p, fu_alt_, f3, hipMemcpyDeviceToDevice, stream_));   // p = r
  launch_cgu_dot(1, g_, ptrs3(fu_), ptrs3(u_r), E, partial_, dscal_ + blk[0], stream_);   // gamma_0 = r:r / N + tiny
  double prev = prev0;  // estimator constructed on the field the step started from
  const bool residual_est = opt_.error_estimator == 1;
  double gamma_cur = 0.0, gamma_0 = 0.0;   // r:r / N + tiny of the current iteration / of the start (residual estimator)
  if (residual_est) {
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotCg + 6, dscal_ + blk[0] + 6, sizeof(double), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    gamma_cur = gamma_0 = hscal_[kSlotCg + 6] / (double)nglobal_ + small;
  }
  long iter = 0;
  bool failed = false;
  bool applied = false;   // u_w = operator(u_p) of the coming iteration is already enqueued
  for (;;) {
    const int cur = (int)(iter & 1), nxt = cur ^ 1;
    if (!applied) {
      if (fused_dir && iter > 0) apply_dir(beta_num, beta_den);                        // p = r + beta p ; u_w = operator(u_p)
      else apply(u_p, Z.v);                                                           // u_w = operator(u_p)
    }
    applied = false;
    if (fused) {
      launch_cgu_tile(0, g_, ptrs3(u_p), ptrs3(fu_alt_), ptrs3(u_p), ptrs3(fu_alt_), ptrs3(fu_cg_), ptrs3(r_alt), Z, dscal_, 0, 0, nvox,
                      small, partial_, dscal_ + s0, stream_);                                         // p : (p - w)
      // eps += alpha p ; r -= alpha (p - w) into the alternate buffers, with the norms of the new eps and r : r
      launch_cgu_tile(1, g_, ptrs3(fu_), ptrs3(u_r), ptrs3(u_p), ptrs3(fu_alt_), ptrs3(fu_cg_), ptrs3(r_alt), E, dscal_, blk[cur] + 6, s0,
                      nvox, small, partial_, dscal_ + blk[nxt], stream_);
      std::swap(fu_, fu_cg_);
      std::swap(u_r, r_alt);
    } else {
    launch_cgu_dot(0, g_, ptrs3(u_p), ptrs3(fu_alt_), Z, partial_, dscal_ + s0, stream_);   // p : (p - w)
    // eps += alpha p ; r -= alpha (p - w),  alpha = gamma / (p:(p - w) / N + tiny)
    launch_cgu_axpy(0, g_, ptrs3(fu_), ptrs3(u_p), ptrs3(u_r), ptrs3(fu_alt_), dscal_, blk[cur] + 6, s0, nvox, small, stream_);
    launch_cgu_dot(1, g_, ptrs3(fu_), ptrs3(u_r), E, partial_, dscal_ + blk[nxt], stream_);   // norms of eps ; r : r
    }
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotCg, dscal_ + blk[nxt], 7 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipMemcpyAsync(herr_, derr_, sizeof(int), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipEventRecord(ev_copy_, stream_));
    if (!cb_ && iter < opt_.maxiter) {
      // p = r + beta p (beta = delta / gamma) and the next operator application, enqueued behind the copies
      if (fused_dir) {
        apply_dir(blk[nxt] + 6, blk[cur] + 6);
      } else {
        launch_cgu_axpy(1, g_, ptrs3(fu_), ptrs3(u_p), ptrs3(u_r), ptrs3(fu_alt_), dscal_, blk[nxt] + 6, blk[cur] + 6, nvox, small,
                        stream_);
        apply(u_p, Z.v);
      }
      applied = true;
    }
    FG_HIP_CHECK(hipEventSynchronize(ev_copy_));
    if (*herr_ != 0) check_device_error("cg");
    // state for accessors called from the callback / bc_error: eps = E + grad_s fu_
    u_valid_ = true;
    eps_stale_ = true;
    for (int c = 0; c < 6; ++c) E_cur_[c] = E.v[c];
    double m[6], s9 = 0.0;
    for (int c = 0; c < 6; ++c) {
      sumsq_[c] = hscal_[kSlotCg + c];
      m[c] = std::sqrt(sumsq_[c] / (double)nglobal_);
    }
    for (int c = 0; c < 6; ++c) s9 += m[c] * m[c];
    for (int c = 3; c < 6; ++c) s9 += m[c] * m[c];
    const double curn = std::sqrt(s9);
    double abs_err = std::fabs(prev - curn);
    double rel_err = abs_err / (small + curn);
    prev = curn;
    if (residual_est) {   // update_cg(gamma, gamma0)  F:14397-14401 with the gamma this iteration started from
      abs_err = std::sqrt(gamma_cur);
      rel_err = std::sqrt(gamma_cur / gamma_0);
      gamma_cur = hscal_[kSlotCg + 6] / (double)nglobal_ + small;   // delta = r:r after the update: the next gamma
    }
    if (std::isnan(rel_err) || cancel_) {  // _converged  F:21177-21244
      failed = true;
      break;
    }
    residuals_.push_back(rel_err);
    if (cb_ && cb_(cb_user_)) break;
    if (cancel_) {
      failed = true;
      break;
    }
    if (iter >= opt_.maxiter) break;
    if (rel_err <= opt_.tol || abs_err <= opt_.abs_tol) {
      double S0[6] = {0, 0, 0, 0, 0, 0};
      if (bc_error(E0, S0) <= opt_.bc_tol) break;
    }
    iter++;
    if (!applied) {   // p = r + beta p
      if (fused_dir) {
        beta_num = blk[nxt] + 6;   // formed inside the next operator application
        beta_den = blk[cur] + 6;
      } else {
        launch_cgu_axpy(1, g_, ptrs3(fu_), ptrs3(u_p), ptrs3(u_r), ptrs3(fu_alt_), dscal_, blk[nxt] + 6, blk[cur] + 6, nvox, small,
                        stream_);
      }
    }
  }
  cg_u_active_ = false;
  iterations_ = iter;
  u_valid_ = true;
  eps_stale_ = true;
  for (int c = 0; c < 6; ++c) E_cur_[c] = E.v[c];
  ensure_eps();
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  solve_time_ += now_seconds() - t_start;
  return failed;
}

// Scalar modes: the same CG in potential space (g = E + grad T_e; r, p, w = grad T_r, T_p, T_w): runCG dispatches every
// non-hyperelastic mode to runCGElasticity (F:22056-22066), whose inner product is the plain sum for 3 components
// (F:20961-20980).  T_e = fu_[0], T_w = fu_alt_[0], T_r / T_p = components 1, 2 of fu_alt_'s buffer mate cg_r_.
bool Solver::run_cg_scalar(const double* E0, double prev0) {
  const double t_start = now_seconds();
  const size_t f1 = (size_t)g_.n * sizeof(double);
  if (!cg_r_) FG_HIP_CHECK(hipMalloc(&cg_r_, 6 * f1));
  double* T_r = cg_r_;
  double* T_p = cg_r_ + g_.n;
  in_run_ = true;
  cg_u_active_ = true;
  const double small = std::numeric_limits<double>::min();
  if (opt_.update_ref) calc_ref_material();
  Vec6 E, Z;
  for (int i = 0; i < 6; ++i) E.v[i] = i < 3 ? E0[i] : 0.0, Z.v[i] = 0.0;
  auto fetch = [&](int slot, int n) {
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + slot, dscal_ + slot, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    check_device_error("cg");
  };
  auto apply = [&](double* T_in, const double* Eadd) {   // T -> f = div((C - C0)(Eadd + grad T)) -> FFT chain -> fu_alt_
    double* keep = fu_;
    fu_ = T_in;
    for (int c = 0; c < 6; ++c) E_cur_[c] = Eadd[c];
    u_pass_front(Eadd);
    fu_ = keep;
    fft_g0_chain(fu_alt_);
  };
  FG_HIP_CHECK(hipMemsetAsync(fu_, 0, f1, stream_));
  apply(fu_, E.v);
  FG_HIP_CHECK(hipMemcpyAsync(T_r, fu_alt_, f1, hipMemcpyDeviceToDevice, stream_));
  FG_HIP_CHECK(hipMemcpyAsync(T_p, fu_alt_, f1, hipMemcpyDeviceToDevice, stream_));
  // Fused form (option cg_fused; as in run_cg_u): p.(p - w) and the update of T_e, T_r with their norms as two tiled sweeps,
  // the direction update inside the operator's sweep, the CG scalars on the device, the next operator application enqueued
  // before the host waits for the sums.  Out of place: T_e, T_r, T_p alternate between two components each (the spare
  // components of fu_ and cg_r_).  Not with a convergence callback (accessors read fu_'s first component in between).
  const int fused_opt = opt_.cg_fused;
  if (fused_opt != 0 && !cb_ && opt_.u_loop >= 2 && sc_sweep_tiled(g_) && !slab_layout_) {
    const int blk[2] = {kSlotCg, kSlotCg + 8}, s0 = kSlotCg + 16;
    const double nvox = (double)nglobal_;
    double *e_cur = fu_, *e_alt = fu_ + g_.n, *r_cur = T_r, *r_alt = cg_r_ + 2 * g_.n, *p_cur = T_p, *p_alt = cg_r_ + 3 * g_.n;
    const double* cond = effective_moduli().p[0];
    auto apply_dir = [&](int i_num, int i_den) {   // T_p := T_r + beta T_p inside the sweep; T_w = operator(T_p)
      time_begin(0);
      launch_sc_sweep_cg(g_, opt_.mu_0, p_cur, r_cur, p_alt, cond, fu_alt_, Z, dscal_, i_num, i_den, nvox, small, partial_,
                         dscal_ + kSlotSumSq, stream_);
      time_end(0);
      fft_g0_chain(fu_alt_);
      std::swap(p_cur, p_alt);
    };
    launch_sc_cg_dot(1, g_, e_cur, r_cur, E, partial_, dscal_ + blk[0], stream_);   // gamma_0 = r.r / N + tiny
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotCg + 6, dscal_ + blk[0] + 6, sizeof(double), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    check_device_error("cg");
    double gamma_cur = hscal_[kSlotCg + 6] / nvox + small;
    const double gamma_0 = gamma_cur;
    double prev = prev0;
    long iter = 0;
    bool failed = false, applied = false;
    for (;;) {
      const int cur = (int)(iter & 1), nxt = cur ^ 1;
      if (!applied) apply(p_cur, Z.v);   // the first iteration: p = r (every later application is enqueued ahead, see below)
      applied = false;
      launch_sc_cgu_tile(0, g_, p_cur, fu_alt_, p_cur, fu_alt_, e_alt, r_alt, Z, dscal_, 0, 0, nvox, small, partial_, dscal_ + s0, stream_);
      launch_sc_cgu_tile(1, g_, e_cur, r_cur, p_cur, fu_alt_, e_alt, r_alt, E, dscal_, blk[cur] + 6, s0, nvox, small, partial_,
                         dscal_ + blk[nxt], stream_);
      std::swap(e_cur, e_alt);
      std::swap(r_cur, r_alt);
      FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotCg, dscal_ + blk[nxt], 7 * sizeof(double), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipMemcpyAsync(herr_, derr_, sizeof(int), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipEventRecord(ev_copy_, stream_));
      if (iter < opt_.maxiter) {   // the next direction and operator application, enqueued behind the copies
        apply_dir(blk[nxt] + 6, blk[cur] + 6);
        applied = true;
      }
      FG_HIP_CHECK(hipEventSynchronize(ev_copy_));
      if (*herr_ != 0) check_device_error("cg");
      double s3 = 0.0;
      for (int c = 0; c < 6; ++c) {
        sumsq_[c] = c < 3 ? hscal_[kSlotCg + c] : 0.0;
        const double m = std::sqrt(sumsq_[c] / nvox);
        s3 += m * m;
      }
      const double curn = std::sqrt(s3);
      double abs_err = std::fabs(prev - curn);
      double rel_err = abs_err / (small + curn);
      prev = curn;
      if (opt_.error_estimator == 1) {   // update_cg(gamma, gamma0)  F:14397-14401 with the gamma this iteration started from
        abs_err = std::sqrt(gamma_cur);
        rel_err = std::sqrt(gamma_cur / gamma_0);
      }
      gamma_cur = hscal_[kSlotCg + 6] / nvox + small;   // delta = r.r after the update: the next gamma
      if (std::isnan(rel_err) || cancel_) {
        failed = true;
        break;
      }
      residuals_.push_back(rel_err);
      if (iter >= opt_.maxiter) break;
      if (rel_err <= opt_.tol || abs_err <= opt_.abs_tol) {
        // bc_error reads the strain of the current iterate: fu_'s first component must be it
        if (e_cur != fu_) {
          FG_HIP_CHECK(hipMemcpyAsync(fu_, e_cur, f1, hipMemcpyDeviceToDevice, stream_));
          std::swap(e_cur, e_alt);
        }
        u_valid_ = true;
        eps_stale_ = true;
        for (int c = 0; c < 6; ++c) E_cur_[c] = E.v[c];
        double S0[6] = {0, 0, 0, 0, 0, 0};
        if (bc_error(E.v, S0) <= opt_.bc_tol) break;
      }
      iter++;
    }
    if (e_cur != fu_) FG_HIP_CHECK(hipMemcpyAsync(fu_, e_cur, f1, hipMemcpyDeviceToDevice, stream_));
    in_run_ = false;
    cg_u_active_ = false;
    iterations_ = iter;
    u_valid_ = true;
    eps_stale_ = true;
    for (int c = 0; c < 6; ++c) E_cur_[c] = E.v[c];
    ensure_eps();
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    solve_time_ += now_seconds() - t_start;
    return failed;
  }
  launch_sc_cg_dot(1, g_, fu_, T_r, E, partial_, dscal_ + kSlotSumSq, stream_);
  fetch(kSlotSumSq, 7);
  double gamma = hscal_[kSlotSumSq + 6] / (double)nglobal_ + small;
  const double gamma_0 = gamma;
  double prev = prev0;
  long iter = 0;
  bool failed = false;
  for (;;) {
    apply(T_p, Z.v);
    launch_sc_cg_dot(0, g_, T_p, fu_alt_, Z, partial_, dscal_ + kSlotMean, stream_);
    fetch(kSlotMean, 1);
    double alpha = hscal_[kSlotMean] / (double)nglobal_ + small;
    alpha = gamma / alpha;
    launch_sc_cg_axpy(0, g_, fu_, T_p, T_r, fu_alt_, alpha, stream_);
    launch_sc_cg_dot(1, g_, fu_, T_r, E, partial_, dscal_ + kSlotSumSq, stream_);
    fetch(kSlotSumSq, 7);
    const double rr = hscal_[kSlotSumSq + 6];
    u_valid_ = true;
    eps_stale_ = true;
    for (int c = 0; c < 6; ++c) E_cur_[c] = E.v[c];
    double s3 = 0.0;
    for (int c = 0; c < 6; ++c) {
      sumsq_[c] = c < 3 ? hscal_[kSlotSumSq + c] : 0.0;
      const double m = std::sqrt(sumsq_[c] / (double)nglobal_);
      s3 += m * m;
    }
    const double cur = std::sqrt(s3);
    double abs_err = std::fabs(prev - cur);
    double rel_err = abs_err / (small + cur);
    prev = cur;
    if (opt_.error_estimator == 1) {   // update_cg(gamma, gamma0)  F:14397-14401
      abs_err = std::sqrt(gamma);
      rel_err = std::sqrt(gamma / gamma_0);
    }
    if (std::isnan(rel_err) || cancel_) {
      failed = true;
      break;
    }
    residuals_.push_back(rel_err);
    if (cb_ && cb_(cb_user_)) break;
    if (cancel_) {
      failed = true;
      break;
    }
    if (iter >= opt_.maxiter) break;
    if (rel_err <= opt_.tol || abs_err <= opt_.abs_tol) {
      double S0[6] = {0, 0, 0, 0, 0, 0};
      if (bc_error(E.v, S0) <= opt_.bc_tol) break;
    }
    iter++;
    const double delta = rr / (double)nglobal_ + small;
    const double beta = delta / gamma;
    gamma = delta;
    launch_sc_cg_axpy(1, g_, fu_, T_p, T_r, fu_alt_, beta, stream_);
  }
  in_run_ = false;
  cg_u_active_ = false;
  iterations_ = iter;
  u_valid_ = true;
  eps_stale_ = true;
  for (int c = 0; c < 6; ++c) E_cur_[c] = E.v[c];
  ensure_eps();
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  solve_time_ += now_seconds() - t_start;
  return failed;
}

bool Solver::run_cg(const double* E0, const double* S0, double prev0) {
  if (nranks_ != 1) throw std::runtime_error("slab-decomposed solvers run method=cg under the slab driver (fg_slab.hip)");
  // (the estimators that measure a mean of the strain field run in strain space, where the iterate is a stored field)
  if (opt_.u_loop >= 2 && u_loop_eligible() && norm2(S0, 6) == 0.0 && opt_.error_estimator < 2) return run_cg_u(E0, prev0);
  const double t_start = now_seconds();
  const size_t f6 = 6 * (size_t)g_.n * sizeof(double);
  for (double** b : {&cg_r_, &cg_p_, &cg_w_})
    if (!*b) FG_HIP_CHECK(hipMalloc(b, f6));
  for (int i = 0; i < 6; ++i) F00_[i] = 0.0;
  u_valid_ = false;
  eps_stale_ = false;
  in_run_ = true;
  const double small = std::numeric_limits<double>::min();
  if (opt_.update_ref) calc_ref_material();
  Vec6 E, Z;
  {
    double t1[6], t2[6], t3[6];  // calcBCMean  F:20242-20245
    voigt_mv(BC_QC0_, E0, t1);
    for (int i = 0; i < 6; ++i) t2[i] = S0[i] - t1[i];
    voigt_mv(BC_M_, t2, t3);
    for (int i = 0; i < 6; ++i) E.v[i] = E0[i] + opt_.bc_relax * t3[i], Z.v[i] = 0.0;
  }
  double prev = prev0;  // estimator constructed on the field the step started from
  auto fetch = [&](int slot, int n) {
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + slot, dscal_ + slot, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    check_device_error("cg");
  };
  const FieldPtrs<6> e = ptrs6(eps_), r = ptrs6(cg_r_), p = ptrs6(cg_p_), w = ptrs6(cg_w_);
  launch_set_const6(g_, e, E, stream_);
  basic_scheme(Z.v, eps_, cg_r_);                                                  // r = -Gamma0 (C - C0) eps
  launch_cg(0, g_, r, e, e, E, 0.0, partial_, dscal_ + kSlotMean, stream_);        // r += E - eps ; r:r
  fetch(kSlotMean, 1);
  double gamma = hscal_[kSlotMean] / (double)nglobal_ + small;
  const double gamma_0 = gamma;
  FG_HIP_CHECK(hipMemcpyAsync(cg_p_, cg_r_, f6, hipMemcpyDeviceToDevice, stream_));  // p = r
  // Round 4 (option cg_fused): the CG scalars on the device, the updates of eps and r as ONE sweep with their norms, the next
  // direction and operator application enqueued before the host waits for the seven sums of the stop rule -- one host
  // synchronisation per iteration instead of three, 528 instead of 576 bytes per voxel of vector work.
  if (opt_.cg_fused != 0) {
    const int blk[2] = {kSlotCg, kSlotCg + 8}, s0 = kSlotCg + 16;
    const double nvox = (double)nglobal_;
    FG_HIP_CHECK(hipMemcpyAsync(dscal_ + blk[0] + 6, dscal_ + kSlotMean, sizeof(double), hipMemcpyDeviceToDevice, stream_));   // gamma_0
    double gamma_cur = gamma;
    long iter = 0;
    bool failed = false, applied = false;
    for (;;) {
      const int cur = (int)(iter & 1), nxt = cur ^ 1;
      if (!applied) basic_scheme(Z.v, cg_p_, cg_w_);                                 // w = -Gamma0 (C - C0) p
      applied = false;
      launch_cg(1, g_, p, w, w, E, 0.0, partial_, dscal_ + s0, stream_);             // p:(p - w)
      // eps += alpha p ; r -= alpha (p - w) ; norms of eps ; r:r
      launch_cg_dev(5, g_, e, r, p, w, dscal_, blk[cur] + 6, s0, nvox, small, partial_, dscal_ + blk[nxt], stream_);
      FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotCg, dscal_ + blk[nxt], 7 * sizeof(double), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipMemcpyAsync(herr_, derr_, sizeof(int), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipEventRecord(ev_copy_, stream_));
      if (!cb_ && iter < opt_.maxiter) {   // p = r + beta p and the next operator application, enqueued behind the copies
        launch_cg_dev(6, g_, e, r, p, w, dscal_, blk[nxt] + 6, blk[cur] + 6, nvox, small, partial_, nullptr, stream_);
        basic_scheme(Z.v, cg_p_, cg_w_);
        applied = true;
      }
      FG_HIP_CHECK(hipEventSynchronize(ev_copy_));
      if (*herr_ != 0) check_device_error("cg");
      double m[6], s9 = 0.0;
      for (int c = 0; c < 6; ++c) {
        sumsq_[c] = hscal_[kSlotCg + c];
        m[c] = std::sqrt(sumsq_[c] / nvox);
      }
      for (int c = 0; c < 6; ++c) s9 += m[c] * m[c];
      for (int c = 3; c < 6; ++c) s9 += m[c] * m[c];
      const double curn = std::sqrt(s9);
      double abs_err = std::fabs(prev - curn);
      double rel_err = abs_err / (small + curn);
      prev = curn;
      if (opt_.error_estimator == 1) {   // update_cg(gamma, gamma0)  F:14397-14401 with the gamma this iteration started from
        abs_err = std::sqrt(gamma_cur);
        rel_err = std::sqrt(gamma_cur / gamma_0);
      }
      gamma_cur = hscal_[kSlotCg + 6] / nvox + small;
      if (opt_.error_estimator >= 2) estimator_update(&abs_err, &rel_err);   // update_cg -> update  F:14465, F:14584
      if (std::isnan(rel_err) || cancel_) {  // _converged  F:21177-21244
        failed = true;
        break;
      }
      residuals_.push_back(rel_err);
      if (cb_ && cb_(cb_user_)) break;
      if (cancel_) {
        failed = true;
        break;
      }
      if (iter >= opt_.maxiter) break;
      if (rel_err <= opt_.tol || abs_err <= opt_.abs_tol) {
        if (bc_error(E0, S0) <= opt_.bc_tol) break;
      }
      iter++;
      if (!applied) launch_cg_dev(6, g_, e, r, p, w, dscal_, blk[nxt] + 6, blk[cur] + 6, nvox, small, partial_, nullptr, stream_);
    }
    in_run_ = false;
    iterations_ = iter;
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    solve_time_ += now_seconds() - t_start;
    return failed;
  }
  long iter = 0;
  bool failed = false;
  for (;;) {
    basic_scheme(Z.v, cg_p_, cg_w_);                                               // w = -Gamma0 (C - C0) p
    launch_cg(1, g_, p, w, w, E, 0.0, partial_, dscal_ + kSlotMean, stream_);      // p:(p - w)
    fetch(kSlotMean, 1);
    double alpha = hscal_[kSlotMean] / (double)nglobal_ + small;
    alpha = gamma / alpha;
    launch_cg(2, g_, e, p, p, E, alpha, partial_, dscal_ + kSlotSumSq, stream_);   // eps += alpha p ; norms
    fetch(kSlotSumSq, 6);
    double m[6], s9 = 0.0;
    for (int c = 0; c < 6; ++c) {
      sumsq_[c] = hscal_[kSlotSumSq + c];
      m[c] = std::sqrt(sumsq_[c] / (double)nglobal_);
    }
    for (int c = 0; c < 6; ++c) s9 += m[c] * m[c];
    for (int c = 3; c < 6; ++c) s9 += m[c] * m[c];
    const double cur = std::sqrt(s9);
    double abs_err = std::fabs(prev - cur);
    double rel_err = abs_err / (small + cur);
    prev = cur;
    if (opt_.error_estimator == 1) {   // update_cg(gamma, gamma0)  F:14397-14401
      abs_err = std::sqrt(gamma);
      rel_err = std::sqrt(gamma / gamma_0);
    }
    if (opt_.error_estimator >= 2) estimator_update(&abs_err, &rel_err);   // update_cg -> update  F:14465, F:14584
    if (std::isnan(rel_err) || cancel_) {  // _converged  F:21177-21244
      failed = true;
      break;
    }
    residuals_.push_back(rel_err);
    if (cb_ && cb_(cb_user_)) break;
    if (cancel_) {
      failed = true;
      break;
    }
    if (iter >= opt_.maxiter) break;
    if (rel_err <= opt_.tol || abs_err <= opt_.abs_tol) {
      if (bc_error(E0, S0) <= opt_.bc_tol) break;
    }
    iter++;
    launch_cg(3, g_, r, p, w, E, -alpha, partial_, dscal_ + kSlotMean, stream_);   // r -= alpha (p - w) ; r:r
    fetch(kSlotMean, 1);
    const double delta = hscal_[kSlotMean] / (double)nglobal_ + small;
    const double beta = delta / gamma;
    gamma = delta;
    launch_cg(4, g_, p, r, r, E, beta, partial_, dscal_ + kSlotMean, stream_);     // p = r + beta p
  }
  in_run_ = false;
  iterations_ = iter;
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
  solve_time_ += now_seconds() - t_start;
  return failed;
}

// ------------------------------------------------------------------ stages and fields
void Solver::run_stage(int stage, const double* E6) {
  FG_HIP_CHECK(hipSetDevice(device_));
  if (opt_.mode == 1) {
    if (stage != kStageIteration) throw std::runtime_error("single stages are not available in heat / porous mode");
    iterate(E6, 1);
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    return;
  }
  ensure_eps();
  if (stage != kStageIteration) u_valid_ = false;  // the stage buffers are being used as scratch
  FieldPtrs<kMaxPhases> phi;
  for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
  FieldPtrs<3> nrm;
  for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
  const double zero6[6] = {0, 0, 0, 0, 0, 0};
  const double* E = E6 ? E6 : zero6;
  switch (stage) {
    case kStageStress:
      if (pt_.n < 1) throw std::runtime_error("No materials specified");
      launch_stress(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(eps_), phi, nrm, ptrs6(tau_), derr_, stream_);
      check_device_error("stress");
      break;
    case kStageStressConst:
      launch_stress_const(g_, opt_.mu_0, opt_.lambda_0, ptrs6(eps_), ptrs6(tau_), stream_);
      break;
    case kStageDiv:
      launch_div(g_, ptrs6(tau_), ptrs3(fu_), XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
      break;
    case kStageFftForward:
      fft_->forward(fu_, 3, g_.n, 1 / (double)nglobal_);
      break;
    case kStageG0: {
      const double alpha = E6 ? E6[0] : -1.0;  // stage tests pass alpha in E6[0]
      const double c10 = -alpha / (opt_.mu_0);
      const double c20 = -alpha / (opt_.mu_0 * (1 + opt_.mu_0 / (opt_.lambda_0 + opt_.mu_0)));
      G0Tables tb;
      for (int a = 0; a < 3; ++a) {
        tb.kpm[a] = g0_kpm_[a];
        tb.kp[a] = g0_kp_[a];
      }
      launch_g0(g_, ptrs3(fu_), tb, c10, c20, G0Layout{0, 0, 0}, stream_);
      break;
    }
    case kStageFftInverse:
      fft_->inverse(fu_, 3, g_.n);
      break;
    case kStageEps: {
      Vec6 Ev, R;
      for (int c = 0; c < 6; ++c) Ev.v[c] = E[c], R.v[c] = 0.0;
      launch_eps_norm(g_, ptrs3(fu_), ptrs6(eps_), Ev, R, false, partial_, dscal_ + kSlotSumSq,
                      XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
      FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotSumSq, dscal_ + kSlotSumSq, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
      FG_HIP_CHECK(hipStreamSynchronize(stream_));
      for (int c = 0; c < 6; ++c) sumsq_[c] = hscal_[kSlotSumSq + c];
      break;
    }
    case kStageIteration:
      basic_scheme(E);
      check_device_error("stress");
      break;
    default:
      throw std::runtime_error("unknown stage");
  }
  FG_HIP_CHECK(hipStreamSynchronize(stream_));
}

int Solver::field_components(const std::string& name) const {
  if (opt_.mode == 1) {
    if (name == "epsilon" || name == "sigma") return 3;
    if (name == "u") return 1;
    if (name == "phi") return pt_.n;
    if (name == "sumsq") return 6;
    return 0;
  }
  if (name == "epsilon" || name == "sigma" || name == "tau") return 6;
  if (name == "u" || name == "f" || name == "normals") return 3;
  if (name == "f_hat") return 3;
  if (name == "phi") return pt_.n;
  if (name == "sumsq") return 6;
  return 0;
}

double* Solver::device_component(const std::string& name, int c) {
  if (name == "epsilon" && c >= 0 && c < 6) return eps_ + (long)c * g_.n;
  if (name == "tau" && c >= 0 && c < 6) return tau_ + (long)c * g_.n;
  if ((name == "f" || name == "u" || name == "f_hat") && c >= 0 && c < 3) return fu_ + (long)c * g_.n;
  if (name == "phi" && c >= 0 && c < pt_.n) return phi_ + (long)c * g_.n;
  if (name == "normals" && normals_ && c >= 0 && c < 3) return normals_ + (long)c * g_.n;
  return nullptr;
}

// get_raw_field  F:15396-15684 (epsilon, sigma, u, phi, normals) + raw stage buffers for the tests
void Solver::get_field(const std::string& name, double* out) {
  FG_HIP_CHECK(hipSetDevice(device_));
  ensure_eps();
  // fu_ is overwritten by the displacement reconstruction (scalar modes: by an equivalent potential, still valid;
  // displacement-space CG: the reconstruction goes to the free buffer fu_alt_, fu_ is the iterate itself)
  if (name == "u" && slab_layout_ && nranks_ > 1)
    throw std::runtime_error("field 'u' (a global reconstruction) is not available on slab-decomposed solvers");
  if (name == "u" && opt_.mode != 1 && !cg_u_active_) u_valid_ = false;
  if (name == "sumsq") {  // the six sums of squares of the last norm sweep (device slot; the displacement loop keeps them there)
    FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotSumSq, dscal_ + kSlotSumSq, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    FG_HIP_CHECK(hipStreamSynchronize(stream_));
    for (int c = 0; c < 6; ++c) out[c] = sumsq_[c] = hscal_[kSlotSumSq + c];
    return;
  }
  if (name == "f_hat") {  // complex [3][nx][ny][nz/2+1], row padding stripped
    download_rows({RowBlock{fu_, out, (long)3 * g_.nx * g_.ny}}, 2 * g_.nzf, 2 * g_.nzc);
    return;
  }
  if (opt_.mode == 1 && name == "sigma") {  // calcStress with C0 = 0  F:15496-15508: the flux
    if (pt_.n < 1) throw std::runtime_error("No materials specified");
    launch_sc_flux(g_, scalar_params(0.0, 1.0), ptrs3(eps_), phase_ptrs(), ptrs3(tau_), stream_);
    download_unpadded(tau_, out, 3, g_.n);
    return;
  }
  if (opt_.mode == 1 && name == "u") {  // potential T = G0 div(C0 : g), alpha = 1  F:15536-15541
    double* const tb = cg_u_active_ ? fu_alt_ : fu_;               // during CG fu_ is the iterate, fu_alt_ is free
    launch_sc_div(g_, ptrs3(eps_), 2 * opt_.mu_0, tb, stream_);    // calcStressConst + divOperatorStaggeredHeat
    const bool timing = timing_;
    timing_ = false;
    fft_g0_chain(tb, 1.0);
    timing_ = timing;
    download_unpadded(tb, out);
    return;
  }
  if (name == "sigma") {  // calcStress with C0 = 0  F:15496-15508
    if (pt_.n < 1) throw std::runtime_error("No materials specified");
    FieldPtrs<kMaxPhases> phi;
    for (int q = 0; q < kMaxPhases; ++q) phi.p[q] = q < pt_.n ? phi_ + (long)q * g_.n : nullptr;
    FieldPtrs<3> nrm;
    for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
    launch_stress(g_, stress_params(0.0, 0.0, 1.0), ptrs6(eps_), phi, nrm, ptrs6(tau_), derr_, stream_);
    check_device_error("sigma");
    download_unpadded(tau_, out, 6, g_.n);
    return;
  }
  if (name == "u" && opt_.mode == 2) {
    // velocity  F:15528-15535: calcStressDiff, div, G0(1/(4 mu0), inf, alpha = 1/(2 mu0)): c10 = c20 = -alpha/mu = -2
    FieldPtrs<kMaxPhases> phi = phase_ptrs();
    FieldPtrs<3> nrm;
    for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
    launch_stress(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(eps_), phi, nrm, ptrs6(tau_), derr_, stream_);
    check_device_error("u");
    launch_div(g_, ptrs6(tau_), ptrs3(fu_), XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
    const double a = 1 / (2 * opt_.mu_0), mu_g = 1 / (4 * opt_.mu_0);
    const double c12[2] = {-a / mu_g, -a / mu_g};
    const bool timing = timing_;
    timing_ = false;
    fft_g0_chain(fu_, a, c12);
    timing_ = timing;
    download_unpadded(fu_, out, 3, g_.n);
    return;
  }
  if (name == "u") {  // u = G0 div (C0 : eps), alpha = 1  F:15509-15521
    double* const ub = cg_u_active_ ? fu_alt_ : fu_;
    launch_stress_const(g_, opt_.mu_0, opt_.lambda_0, ptrs6(eps_), ptrs6(tau_), stream_);
    launch_div(g_, ptrs6(tau_), ptrs3(ub), XHalo{{nullptr, nullptr}, {nullptr, nullptr}}, stream_);
    fft_->forward(ub, 3, g_.n, 1 / (double)nglobal_);
    const double alpha = 1.0;
    const double c10 = -alpha / (opt_.mu_0);
    const double c20 = -alpha / (opt_.mu_0 * (1 + opt_.mu_0 / (opt_.lambda_0 + opt_.mu_0)));
    G0Tables tb;
    for (int a = 0; a < 3; ++a) {
      tb.kpm[a] = g0_kpm_[a];
      tb.kp[a] = g0_kp_[a];
    }
    launch_g0(g_, ptrs3(ub), tb, c10, c20, G0Layout{0, 0, 0}, stream_);
    fft_->inverse(ub, 3, g_.n);
    download_unpadded(ub, out, 3, g_.n);
    return;
  }
  const int nc = field_components(name);
  if (nc == 0) throw std::runtime_error("Unknown field '" + name + "'");
  std::vector<RowBlock> blocks;
  for (int c = 0; c < nc; ++c) {
    double* d = device_component(name, c);
    if (!d) throw std::runtime_error("field '" + name + "' is not available");
    blocks.push_back(RowBlock{d, out + (long)c * g_.nxyz, (long)g_.nx * g_.ny});
  }
  download_rows(blocks, g_.nz, g_.nzp);
}

void Solver::set_field(const std::string& name, const double* in) {
  FG_HIP_CHECK(hipSetDevice(device_));
  if (opt_.mode == 1) throw std::runtime_error("fields cannot be set in heat / porous mode");
  ensure_eps();
  u_valid_ = false;
  su_valid_ = false;   // slab driver: the strain field is the state again
  if (name == "f_hat") {
    upload_rows({RowBlock{fu_, const_cast<double*>(in), (long)3 * g_.nx * g_.ny}}, 2 * g_.nzf, 2 * g_.nzc);
    return;
  }
  if (name == "normals") {
    set_normals(in);
    return;
  }
  const int nc = field_components(name);
  if (nc == 0 || name == "sigma" || name == "sumsq") throw std::runtime_error("field '" + name + "' cannot be set");
  std::vector<RowBlock> blocks;
  for (int c = 0; c < nc; ++c) {
    double* d = device_component(name, c);
    if (!d) throw std::runtime_error("field '" + name + "' is not available");
    blocks.push_back(RowBlock{d, const_cast<double*>(in) + (long)c * g_.nxyz, (long)g_.nx * g_.ny});
  }
  upload_rows(blocks, g_.nz, g_.nzp);
}

}  // namespace fg
