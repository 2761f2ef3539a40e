// Slab-decomposed Lippmann-Schwinger loop: per-member steps (Solver::slab_*) and the collective driver (SlabGroup).
// See fg_slab.h, fg_slab_plan.h (layouts, exchange plan) and fg_comm.h (transports).
#include "fg_slab.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

#include "fg_hip_util.h"
#include "fg_slab_plan.h"

namespace fg {

using namespace hostmath;
using namespace slots;

namespace {
constexpr double kEps = 2.220446049250313e-16;
// comm -> compute event slots
constexpr int kXA2AFwd = 0, kXA2ABwd = 3, kXHaloU = 6, kXHaloTau = 7, kXModuli = 8, kXSums = 9;

double now_seconds() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}

// y lengths the blocked FFT pass does not cover: plain [nxl][ny][nzc] <-> blocked [q][nxl][nyl][nzc] copy of one component
__global__ void k_block_remap(const cplx* in, cplx* out, int nxl, int ny, int nyl, int nzc, int to_blocked) {
  const long total = (long)nxl * ny * nzc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / nzc;
    const int k = (int)(idx - row * nzc);
    const int x = (int)(row / ny);
    const int j = (int)(row - (long)x * ny);
    const int q = j / nyl, jl = j - q * nyl;
    const long b = (long)q * nxl * nyl * nzc + ((long)x * nyl + jl) * nzc + k;
    if (to_blocked) out[b] = in[idx];
    else out[idx] = in[b];
  }
}

void launch_block_remap(const double* in, double* out, const Grid& g, int nyl, bool to_blocked, hipStream_t s) {
  const long total = (long)g.nx * g.ny * g.nzc;
  long nb = (total + 255) / 256;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(k_block_remap, dim3((unsigned)nb), dim3(256), 0, s, reinterpret_cast<const cplx*>(in),
                     reinterpret_cast<cplx*>(out), g.nx, g.ny, nyl, g.nzc, to_blocked ? 1 : 0);
  FG_HIP_CHECK(hipGetLastError());
}

// The flag word every reduction of the driver carries (summed over the ranks): whether a kernel of this rank raised the
// device error flag, and whether this rank was asked to stop.  Stop and error decisions are taken on the reduced word, so
// every rank takes the same one.
__global__ void k_flag_word(const int* derr, double stop, double* flag) {
  flag[0] = *derr != 0 ? 1.0 : 0.0;
  flag[1] = stop;
}

FieldPtrs<3> strided3(double* base, long stride) {
  FieldPtrs<3> f;
  for (int c = 0; c < 3; ++c) f.p[c] = base + c * stride;
  return f;
}
}  // namespace

// ===================================================================================================== member side
void Solver::connect(std::unique_ptr<Comm> comm, std::shared_ptr<SlabGroup> group) {
  if (!slab_layout_) throw std::runtime_error("not a slab solver (create it with fg_create_slab or fg_slab_group_create)");
  if (comm_ || group_) throw std::runtime_error("slab solver is already connected to a transport");
  if (comm && (comm->size() != nranks_ || comm->rank() != rank_))
    throw std::runtime_error("transport rank / size differ from the solver's");
  FG_HIP_CHECK(hipSetDevice(device_));
  comm_ = std::move(comm);
  group_ = std::move(group);
  if (comm_ && std::strcmp(comm_->name(), "rccl") == 0) {
    // exchanges overlap the transforms of the next component: own stream, joined by events
    FG_HIP_CHECK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
    owns_comm_stream_ = true;
  }
}

SlabGroup& Solver::slab_group() {
  if (!slab_layout_) throw std::runtime_error("not a slab solver");
  if (!group_) {
    if (nranks_ != 1) throw std::runtime_error("slab solver is not connected to a transport (fg_slab_connect_*)");
    group_ = std::make_shared<SlabGroup>(std::vector<Solver*>{this});   // a lone slab is periodic in itself
  }
  return *group_;
}

void Solver::slab_alloc() {
  if (su_[0]) return;
  FG_HIP_CHECK(hipSetDevice(device_));
  ucs_ = g_.n + 4 * g_.nyzp;
  for (int k = 0; k < 2; ++k) {
    FG_HIP_CHECK(hipMalloc(&su_[k], 3 * (size_t)ucs_ * sizeof(double)));
    FG_HIP_CHECK(hipMemsetAsync(su_[k], 0, 3 * (size_t)ucs_ * sizeof(double), stream_));
  }
  FG_HIP_CHECK(hipMalloc(&smod_, 2 * (size_t)ucs_ * sizeof(double)));
  FG_HIP_CHECK(hipMemsetAsync(smod_, 0, 2 * (size_t)ucs_ * sizeof(double), stream_));
  smod_dirty_ = true;
  gu_ = g_;
  gu_.xw_lo = g_.nx + 4;   // plane -1 -> spare plane nx + 3
  gu_.xw_hi = 0;           // planes nx, nx + 1, nx + 2 -> themselves (the last two are loaded by the march, never used)
  fft_ys_.reset(new Fft3(make_grid(nxg_, nyl_, g_.nz, 1.0, 1.0, 1.0), stream_));
  FG_HIP_CHECK(hipEventCreateWithFlags(&ev_c2x_, hipEventDisableTiming));
  FG_HIP_CHECK(hipEventCreateWithFlags(&ev_norm_, hipEventDisableTiming));
  for (int k = 0; k < kCommSlots; ++k) FG_HIP_CHECK(hipEventCreateWithFlags(&ev_x_[k], hipEventDisableTiming));
  for (int k = 0; k < 2; ++k) FG_HIP_CHECK(hipEventCreate(&ev_ct_[k]));
}

void Solver::comm_begin() {
  if (comm_stream_ == stream_) return;
  FG_HIP_CHECK(hipEventRecord(ev_c2x_, stream_));
  FG_HIP_CHECK(hipStreamWaitEvent(comm_stream_, ev_c2x_, 0));
}
void Solver::comm_end(int slot) {
  if (comm_stream_ == stream_) return;
  FG_HIP_CHECK(hipEventRecord(ev_x_[slot], comm_stream_));
  x_pending_[slot] = true;
}
void Solver::comm_wait(int slot) {
  if (!x_pending_[slot]) return;
  FG_HIP_CHECK(hipStreamWaitEvent(stream_, ev_x_[slot], 0));
  x_pending_[slot] = false;
}

// stage timing: HIP events on the exchange stream around one exchange; the host waits for it (every rank does, in the same
// order), so the figure is the exchange alone -- transfer plus the wait for the slowest peer -- with nothing overlapped
void Solver::comm_time_begin() {
  if (timing_) FG_HIP_CHECK(hipEventRecord(ev_ct_[0], comm_stream_));
}
void Solver::comm_time_end(int category) {
  if (!timing_) return;
  FG_HIP_CHECK(hipEventRecord(ev_ct_[1], comm_stream_));
  FG_HIP_CHECK(hipEventSynchronize(ev_ct_[1]));
  float ms = 0.f;
  FG_HIP_CHECK(hipEventElapsedTime(&ms, ev_ct_[0], ev_ct_[1]));
  comm_ms_[category] += ms;
}

// One slab has nothing to overlap; otherwise split when a component of the slab is >= 32 MB (512^3 on 8 GPUs: 135 MB,
// 256^3 on 8: 17 MB).  Option slab_split = 0 / 1 overrides (tests, A/B runs).
bool Solver::slab_split() const {
  if (opt_.slab_split >= 0) return opt_.slab_split != 0;
  return (nranks_ > 1 || slab_loopback()) && (double)g_.n * sizeof(double) >= 32.0 * 1024 * 1024;
}

// Batched exchanges (no split) with the three components of a peer in one message: needs the y pass that writes / reads the
// blocked layout and the radix fused x pass (powers of two all round).  Option slab_interleave = 0 keeps
// one message per peer and component.
bool Solver::slab_interleave() const {
  if (opt_.slab_interleave == 0) return false;
  if (slab_split() || !(nranks_ > 1 || slab_loopback()) || opt_.mode != 0) return false;
  const int nxl = g_.nx;
  return fft_ && fft_ys_ && fft_->can_block_y(nranks_) && opt_.fuse_x && nxg_ > 1 && nxg_ <= 512 && fft_ys_->fast_x() &&
         fft_ys_->can_fuse(0) && nxl > 0 && (nxl & (nxl - 1)) == 0;
}

// Test mode for boxes with ONE GPU: a lone slab connected to a transport sends its all-to-all blocks and halo planes to
// itself THROUGH the transport (RCCL: ncclSend / ncclRecv to the own rank inside a group, ncclAllReduce over one rank), on
// the second stream with the event choreography of the multi-GPU run.
bool Solver::slab_loopback() const { return nranks_ == 1 && comm_ && opt_.slab_loopback != 0; }

double* Solver::slab_buffer(int id) {
  switch (id) {
    case FG_BUF_SPECTRUM_X: return tau_;
    // one slab: the identity.  Otherwise the y-slab spectrum lands in fu_: f has been consumed by the forward y pass when the
    // all-to-all of its component is posted, and the next sweep writes fu_ only behind the halo exchange that follows the
    // backward all-to-all -- one field less in the working set of a pass (256^3 on 8 ranks: 291 -> 239 MB, inside the 256 MB
    // Infinity Cache).  Valid for receiver-gated transports only (Comm::exchange's contract); one that pushes gets tau_ + 3 n.
    case FG_BUF_SPECTRUM_Y:
      if (nranks_ == 1 && !slab_loopback()) return tau_;
      return (comm_ && comm_->pushes()) ? tau_ + 3 * g_.n : fu_;
    case FG_BUF_U: return su_[su_cur_ ^ 1];                                  // the displacement the chain is producing
    case FG_BUF_MODULI: return smod_;
    case FG_BUF_HALO_SEND_LO: return halo_[0];
    case FG_BUF_HALO_SEND_HI: return halo_[1];
    case FG_BUF_HALO_RECV_LO: return halo_[2];
    case FG_BUF_HALO_RECV_HI: return halo_[3];
  }
  throw std::runtime_error("unknown slab buffer");
}

// One exchange of the plan.  A lone slab is periodic in itself: its halo planes are its own planes.
void Solver::slab_exchange(int what, int comp, int done_slot) {
  SlabDims d = slab_dims(nxg_, g_.ny, g_.nz, nranks_, rank_);
  d.loopback = slab_loopback();
  d.ncomp_u = opt_.mode == 1 ? 1 : 3;
  const size_t pb = (size_t)d.plane * sizeof(double);
  if (nranks_ == 1 && !d.loopback) {
    if (what == FG_PLAN_HALO_U || what == FG_PLAN_HALO_MODULI) {
      double* base = slab_buffer(what == FG_PLAN_HALO_U ? FG_BUF_U : FG_BUF_MODULI);
      const int nc = what == FG_PLAN_HALO_U ? d.ncomp_u : 2;
      for (int c = 0; c < nc; ++c) {
        double* b = base + c * d.ucs;
        FG_HIP_CHECK(hipMemcpyAsync(b + slab_hi_plane(d) * d.plane, b, pb, hipMemcpyDeviceToDevice, stream_));
        FG_HIP_CHECK(hipMemcpyAsync(b + slab_lo_plane(d) * d.plane, b + (long)(d.nxl - 1) * d.plane, pb, hipMemcpyDeviceToDevice,
                                    stream_));
      }
    } else if (what == FG_PLAN_HALO_TAU) {
      FG_HIP_CHECK(hipMemcpyAsync(halo_[2], halo_[1], pb, hipMemcpyDeviceToDevice, stream_));
      FG_HIP_CHECK(hipMemcpyAsync(halo_[3], halo_[0], 2 * pb, hipMemcpyDeviceToDevice, stream_));
    }
    return;
  }
  if (!comm_) throw std::runtime_error("slab solver is not connected to a transport (fg_slab_connect_*)");
  // comp < 0: the three components of an all-to-all in ONE exchange (small slabs: fewer, larger messages) -- one message
  // per peer where the interleaved layout is available, else one per peer and component
  std::vector<XOp> ops;
  comm_begin();
  const bool one_msg = comp < 0 && slab_interleave() && (what == FG_PLAN_A2A_FORWARD || what == FG_PLAN_A2A_BACKWARD);
  for (int c = comp < 0 ? (one_msg ? -1 : 0) : comp; c <= (comp < 0 ? (one_msg ? -1 : 2) : comp); ++c) {
    const SlabPlan p = slab_plan(d, what, c);
    for (const fg_plan_op& o : p.ops)
      ops.push_back(XOp{o.send, o.peer, slab_buffer(o.buffer) + o.offset, (size_t)o.count * sizeof(double)});
    if (p.self_src.count)
      FG_HIP_CHECK(hipMemcpyAsync(slab_buffer(p.self_dst.buffer) + p.self_dst.offset,
                                  slab_buffer(p.self_src.buffer) + p.self_src.offset, (size_t)p.self_src.count * sizeof(double),
                                  hipMemcpyDeviceToDevice, comm_stream_));
  }
  comm_time_begin();
  comm_->exchange(ops.data(), (int)ops.size(), comm_stream_);
  comm_time_end(what == FG_PLAN_A2A_FORWARD ? 0 : what == FG_PLAN_A2A_BACKWARD ? 1 : 2);
  comm_end(done_slot);
}

// All-reduce of dscal_[slot .. slot + n) together with the flag word.  The exchange stream is ordered behind the compute
// stream here in every configuration (also for a lone slab, whose sums need no reduction): the copies to the host that
// follow are issued on the exchange stream.
void Solver::slab_reduce(int slot, int n, bool min_op) {
  hipLaunchKernelGGL(k_flag_word, dim3(1), dim3(1), 0, stream_, derr_, cancel_ ? 1.0 : 0.0, dscal_ + kSlotFlag);
  FG_HIP_CHECK(hipGetLastError());
  comm_begin();
  if (nranks_ > 1 || slab_loopback()) {
    if (!comm_) throw std::runtime_error("slab solver is not connected to a transport (fg_slab_connect_*)");
    comm_time_begin();
    comm_->group_begin();
    comm_->allreduce(dscal_ + slot, n, min_op, comm_stream_);
    comm_->allreduce(dscal_ + kSlotFlag, 2, false, comm_stream_);
    comm_->group_end();
    comm_time_end(3);
  }
  comm_end(kXSums);
}

void Solver::slab_fetch_norms(int n) {
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotSumSq, dscal_ + kSlotSumSq, n * sizeof(double), hipMemcpyDeviceToHost, comm_stream_));
  FG_HIP_CHECK(hipMemcpyAsync(hscal_ + kSlotFlag, dscal_ + kSlotFlag, 2 * sizeof(double), hipMemcpyDeviceToHost, comm_stream_));
  FG_HIP_CHECK(hipMemcpyAsync(herr_, derr_, sizeof(int), hipMemcpyDeviceToHost, comm_stream_));
  FG_HIP_CHECK(hipEventRecord(ev_norm_, comm_stream_));
}

bool Solver::slab_fast_ok(bool allow_mixed_bc) const {
  if (opt_.mode == 1)   // heat / porous: the potential sweep on the precomputed conductivity
    return opt_.u_loop >= 2 && opt_.gamma_scheme == 0 && pt_.n >= 1 && opt_.mixing == kMixVoigt && opt_.bc_relax == 1.0 && fft_ys_ &&
           (frobenius(BC_MQ_) < kEps || allow_mixed_bc);
  if (!(opt_.u_loop >= 2 && opt_.u_tile && opt_.mode == 0 && opt_.gamma_scheme == 0 && pt_.n >= 1 &&
        (opt_.mixing == kMixVoigt || (opt_.mixing == kMixLaminate && normals_)) && opt_.bc_relax == 1.0 && u_tile_supported(g_)))
    return false;
  return frobenius(BC_MQ_) < kEps || allow_mixed_bc;
}

void Solver::slab_moduli_step() {
  if (!smod_dirty_) return;
  FieldPtrs<2> mod;
  mod.p[0] = smod_;
  mod.p[1] = smod_ + ucs_;
  // two complementary phases: the sweep reads phi_1 (with its halo planes) and forms the moduli itself.  The decision is
  // local data; every rank of a run sees the same kind of phase fields (normalizePhi output or not)
  slab_phi_ = opt_.mode == 0 && two_phase_complementary();
  if (slab_phi_) {
    FG_HIP_CHECK(hipMemcpyAsync(smod_, phi_ + g_.n, (size_t)g_.n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  } else {
    PhaseTable t = phase_table();
    if (opt_.mode == 1)   // scalar modes: k_effective_moduli stores sum phi 2 mu, the sweep wants the conductivity sum phi mu
      for (int q = 0; q < kMaxPhases; ++q) t.mu[q] = 0.5 * pt_.mu[q], t.lambda[q] = 0.0;
    launch_effective_moduli(g_, t, phase_ptrs(), mod, stream_);
  }
  slab_exchange(FG_PLAN_HALO_MODULI, 0, kXModuli);
  smod_dirty_ = false;
}

// u_k (with the +-1 planes of the neighbours) -> all-reduced sums of squares of eps_k (, sums of tau), f_{k+1} in fu_
void Solver::slab_front_fast(const double* E6, bool sum_tau, const double* u_src, bool reduce) {
  double* u_in = u_src ? const_cast<double*>(u_src) : su_[su_cur_];
  comm_wait(kXHaloU);
  comm_wait(kXModuli);
  comm_wait(kXSums);
  Vec6 E;
  for (int c = 0; c < 6; ++c) E.v[c] = E6[c];
  FieldPtrs<2> mod;
  mod.p[0] = smod_;
  mod.p[1] = smod_ + ucs_;
  time_begin(0);
  if (opt_.mode == 1) {
    // heat / porous: T_k (component 0 of the displacement buffer, with its halo planes) -> sums of squares of g_k = E + grad T_k,
    // f = div((a - 2 mu0) g_k)
    // (mixed BC: the sweep also leaves the three sums of the flux polarisation in kSlotMean; entries 3..5 stay zero)
    if (sum_tau) FG_HIP_CHECK(hipMemsetAsync(dscal_ + kSlotMean, 0, 6 * sizeof(double), stream_));
    const bool have_sums = launch_sc_sweep_fast(gu_, opt_.mu_0, u_in, smod_, fu_, E, partial_, dscal_ + kSlotSumSq, stream_,
                                                sum_tau ? dscal_ + kSlotMean : nullptr);
    if (sum_tau && !have_sums) {
      // grids the tiled sweep does not fit: sums of tau = P(g) - 2 mu0 g from the gradient field (two more sweeps, as on one GPU)
      launch_sc_grad(gu_, u_in, ptrs3(eps_), E, partial_, dscal_ + kSlotScratch, stream_);
      launch_sc_flux_mean(g_, scalar_params(opt_.mu_0, 1.0), ptrs3(eps_), phase_ptrs(), partial_, dscal_ + kSlotMean, stream_);
    }
    time_end(0);
    if (reduce) slab_reduce(kSlotSumSq, sum_tau ? 12 : 6, false);
    return;
  }
  const bool laminate = opt_.mixing == kMixLaminate;
  if (laminate) {
    // laminate mixing = the Voigt sweep + the divergence of d = tau_laminate - tau_voigt on the interface voxels (see
    // k_interface_strain).  d of the slab's two boundary planes also enters the divergence on the neighbouring slabs: it
    // travels as dense planes (the polarisation halo of the strain-state pipeline) while the sweep runs.
    build_laminate_lists();
    launch_interface_delta(gu_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), strided3(u_in, ucs_), E, mixed_list_, mixed_n_,
                           lam_epsc_, lam_phic_, lam_nrmc_, dtau_, derr_, stream_);
    launch_delta_pack(g_, mixed_list_, mixed_n_, dtau_, halo_[0], halo_[1], stream_);
    slab_exchange(FG_PLAN_HALO_TAU, 0, kXHaloTau);
  }
  const PhaseTable pt2 = phase_table();
  launch_u_tile(gu_, opt_.mu_0, opt_.lambda_0, strided3(u_in, ucs_), mod, ptrs3(fu_), E, partial_, dscal_ + kSlotSumSq,
                stream_, sum_tau, slab_phi_ ? &pt2 : nullptr);
  if (laminate) return;   // slab_front_laminate (next step: the planes of the neighbours have to be posted first)
  time_end(0);
  if (reduce) slab_reduce(kSlotSumSq, sum_tau ? 12 : 6, false);
}

// second step of the sweep with laminate mixing: div d of the own interface voxels and of the neighbours' boundary planes
void Solver::slab_front_laminate(bool sum_tau, bool reduce) {
  if (opt_.mixing != kMixLaminate || opt_.mode != 0) return;
  launch_delta_div(g_, aff_list_, aff_slots_, aff_n_, dtau_, ptrs3(fu_), stream_);
  comm_wait(kXHaloTau);
  launch_delta_div_halo(g_, halo_[2], halo_[3], ptrs3(fu_), stream_);
  if (sum_tau) {   // mixed BC: <tau_laminate> = <tau_voigt> (from the sweep) + sum of the differences / N
    launch_sum_dtau(dtau_, mixed_n_, partial_, dscal_ + kSlotScratch, stream_);
    launch_add_small(dscal_ + kSlotMean, dscal_ + kSlotScratch, 6, stream_);
  }
  time_end(0);
  if (reduce) slab_reduce(kSlotSumSq, sum_tau ? 12 : 6, false);
}

void Solver::slab_cg_alloc() {
  if (scg_) return;
  FG_HIP_CHECK(hipSetDevice(device_));
  FG_HIP_CHECK(hipMalloc(&scg_, 6 * (size_t)ucs_ * sizeof(double)));
  FG_HIP_CHECK(hipMemsetAsync(scg_, 0, 6 * (size_t)ucs_ * sizeof(double), stream_));
}

bool Solver::slab_cg_alloc_fused() {
  FG_HIP_CHECK(hipSetDevice(device_));
  const size_t f3 = 3 * (size_t)ucs_ * sizeof(double);
  if (!scg2_ || !su_alt_) {
    size_t free_b = 0, total_b = 0;
    FG_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    if (free_b < (scg2_ ? 0 : 2 * f3) + (su_alt_ ? 0 : f3) + (size_t)(0.02 * (double)total_b)) return false;
    if (!scg2_) FG_HIP_CHECK(hipMalloc(&scg2_, 2 * f3));
    if (!su_alt_) FG_HIP_CHECK(hipMalloc(&su_alt_, f3));
    FG_HIP_CHECK(hipMemsetAsync(scg2_, 0, 2 * f3, stream_));
    FG_HIP_CHECK(hipMemsetAsync(su_alt_, 0, f3, stream_));
  }
  return true;
}

void Solver::slab_front_fast_cg(const double* E6, int i_num, int i_den, double nvox, double small) {
  comm_wait(kXHaloU);
  comm_wait(kXModuli);
  comm_wait(kXSums);
  Vec6 E;
  for (int c = 0; c < 6; ++c) E.v[c] = E6[c];
  FieldPtrs<2> mod;
  mod.p[0] = smod_;
  mod.p[1] = smod_ + ucs_;
  const PhaseTable pt2 = phase_table();
  time_begin(0);
  launch_u_tile_cg(gu_, opt_.mu_0, opt_.lambda_0, strided3(cgs_p_, ucs_), strided3(cgs_r_, ucs_), strided3(cgs_pa_, ucs_), mod, ptrs3(fu_),
                   E, dscal_, i_num, i_den, nvox, small, partial_, dscal_ + kSlotSumSq, stream_, slab_phi_ ? &pt2 : nullptr);
  // the spare planes of the new direction, point-wise (its halo planes stay valid without an exchange)
  launch_cgu_axpy_oop(1, strided3(cgs_p_, ucs_), strided3(cgs_p_, ucs_), strided3(cgs_r_, ucs_), strided3(cgs_p_, ucs_),
                      strided3(cgs_pa_, ucs_), strided3(cgs_pa_, ucs_), dscal_, i_num, i_den, nvox, small, g_.n, ucs_ - g_.n, stream_);
  time_end(0);
  std::swap(cgs_p_, cgs_pa_);
}

void Solver::slab_front_fast_sc_cg(int i_num, int i_den, double nvox, double small) {
  comm_wait(kXHaloU);
  comm_wait(kXModuli);
  comm_wait(kXSums);
  Vec6 Z;
  for (int c = 0; c < 6; ++c) Z.v[c] = 0.0;
  time_begin(0);
  launch_sc_sweep_cg(gu_, opt_.mu_0, cgs_p_, cgs_r_, cgs_pa_, smod_, fu_, Z, dscal_, i_num, i_den, nvox, small, partial_,
                     dscal_ + kSlotSumSq, stream_);
  launch_sc_cg_axpy_oop(1, cgs_p_, cgs_p_, cgs_r_, cgs_p_, cgs_pa_, cgs_pa_, dscal_, i_num, i_den, nvox, small, g_.n, ucs_ - g_.n, stream_);
  time_end(0);
  std::swap(cgs_p_, cgs_pa_);
}

// The transform chain of one pass, cut into steps that each end in one exchange (k = 1..9):
//   1..3  component c = k-1: z r2c, y c2c into the all-to-all layout          | all-to-all(c) forward
//   4     x c2c + Green operator + x c2c^-1 on the y-slab, three components   | all-to-all(0) back
//   5, 6                                                                      | all-to-all(1), (2) back
//   7..9  component c = k-7: y c2c^-1 out of the all-to-all layout, z c2r     | (9:) halo planes of the new u
// f is read from fu_, the new displacement lands in su_[next].
void Solver::slab_chain_step(int k) {
  const long n = g_.n;
  double* S = slab_buffer(FG_BUF_SPECTRUM_X);
  double* R = slab_buffer(FG_BUF_SPECTRUM_Y);
  double* un = su_[su_cur_ ^ 1];
  const bool blocked = fft_->can_block_y(nranks_);
  // Large slabs are bandwidth bound on the links: component c's all-to-all overlaps the transforms of c + 1.  Small
  // slabs are latency bound (launches, RCCL start-up): the three components go through every stage together.
  const int NC = opt_.mode == 1 ? 1 : 3;   // heat / porous: one potential
  const bool split = slab_split() && NC == 3;
  const bool inter = slab_interleave();   // batched mode, one message per peer: blocks of three components
  const SlabDims sd = slab_dims(nxg_, g_.ny, g_.nz, nranks_, rank_);
  auto forward = [&](int c0, int nc) {   // z r2c and y c2c (into the all-to-all layout) of components c0 .. c0 + nc - 1
    double* f = fu_ + c0 * n;
    time_begin(2);
    fft_->r2c_z(f, nc, n);
    time_end(2);
    time_begin(3);
    if (inter) {
      fft_->c2c_y_blocked(f, n, S, sd.block, nc, -1, 1.0, nranks_, 3);
    } else if (blocked) {
      fft_->c2c_y_blocked(f, n, S + c0 * n, n, nc, -1, 1.0, nranks_);
    } else {
      fft_->c2c_y(f, nc, n, -1, 1.0);
      for (int c = c0; c < c0 + nc; ++c) launch_block_remap(fu_ + c * n, S + c * n, g_, nyl_, true, stream_);
    }
    time_end(3);
  };
  auto backward = [&](int c0, int nc) {  // y c2c^-1 (out of the all-to-all layout) and z c2r
    double* u = un + c0 * ucs_;
    time_begin(7);
    if (inter) {
      fft_->c2c_y_blocked(S, sd.block, u, ucs_, nc, +1, 1.0, nranks_, 3);
    } else if (blocked) {
      fft_->c2c_y_blocked(S + c0 * n, n, u, ucs_, nc, +1, 1.0, nranks_);
    } else {
      for (int c = c0; c < c0 + nc; ++c) launch_block_remap(S + c * n, un + c * ucs_, g_, nyl_, false, stream_);
      fft_->c2c_y(u, nc, ucs_, +1, 1.0);
    }
    time_end(7);
    time_begin(8);
    fft_->c2r_z(u, nc, ucs_);
    time_end(8);
  };
  auto finish = [&]() {
    slab_exchange(FG_PLAN_HALO_U, 0, kXHaloU);
    if (timing_) times_.count++;
  };
  if (k >= 1 && k <= 3) {
    if (split) {
      forward(k - 1, 1);
      slab_exchange(FG_PLAN_A2A_FORWARD, k - 1, kXA2AFwd + k - 1);
    } else if (k == 1) {
      forward(0, NC);
      slab_exchange(FG_PLAN_A2A_FORWARD, NC == 1 ? 0 : -1, kXA2AFwd);
    }
  } else if (k == 4) {
    for (int c = 0; c < 3; ++c) comm_wait(kXA2AFwd + c);
    const double alpha = -1.0;   // GammaOperator(..., -1)  F:20575
    const double scale = 1 / (double)nglobal_;
    G0Params gp;
    gp.c10 = -alpha / (opt_.mu_0);
    gp.c20 = -alpha / (opt_.mu_0 * (1 + opt_.mu_0 / (opt_.lambda_0 + opt_.mu_0)));
    if (opt_.mode == 2)   // dual Stokes scheme: Green operator of mu = -mu0 (= -1/(4 m)), lambda = inf: c20 = c10  (F:20444, F:19749-19755)
      gp.c10 = gp.c20 = alpha / opt_.mu_0;
    gp.inv_h0 = 2.0 * nxg_ / g_.dx;
    G0Tables tb;
    for (int a = 0; a < 3; ++a) {
      tb.kpm[a] = gp.kpm[a] = g0_kpm_[a];
      tb.kp[a] = gp.kp[a] = g0_kp_[a];
    }
    const int jj0 = rank_ * nyl_;   // this rank's ky rows
    time_begin(5);
    if (NC == 1) {
      // G0OperatorFourierStaggeredHeat  F:19758-19823: c1 = c10 / |k|^2, c10 = -alpha / (2 mu0)
      gp.c10 = -alpha / (2 * opt_.mu_0);
      gp.c20 = 0.0;
      if (opt_.fuse_x && nxg_ > 1 && fft_ys_->can_fuse(0, 1)) {
        fft_ys_->fused_g0(R, n, 0, scale, gp, jj0, 1);   // inside the fused x pass
      } else {
        if (nxg_ > 1) fft_ys_->c2c_x(R, 1, n, -1, scale);
        else fft_ys_->scale(R, 1, n, scale);
        launch_g0_heat(make_grid(nxg_, nyl_, g_.nz, 1.0, 1.0, 1.0), R, tb, gp.c10, stream_, jj0);
        if (nxg_ > 1) fft_ys_->c2c_x(R, 1, n, +1, 1.0);
      }
    } else if (inter) {
      // y-slab [p][c][nxl][nyl][nzc]: component stride one block, x plane j at j * ls + (j / nxl) * 2 blocks
      int sh = 0;
      while ((1 << sh) < g_.nx) ++sh;
      fft_ys_->fused_g0(R, sd.block, 0, scale, gp, jj0, 3, sh, 2 * sd.block);
    } else if (opt_.fuse_x && nxg_ > 1 && fft_ys_->can_fuse(0)) {
      fft_ys_->fused_g0(R, n, 0, scale, gp, jj0);
    } else {
      if (nxg_ > 1) fft_ys_->c2c_x(R, 3, n, -1, scale);
      else fft_ys_->scale(R, 3, n, scale);
      Grid gy = make_grid(nxg_, nyl_, g_.nz, 1.0, 1.0, 1.0);
      launch_g0(gy, ptrs3(R), tb, gp.c10, gp.c20, G0Layout{0, nyl_, jj0}, stream_);
      if (nxg_ > 1) fft_ys_->c2c_x(R, 3, n, +1, 1.0);
    }
    time_end(5);
    slab_exchange(FG_PLAN_A2A_BACKWARD, (split || NC == 1) ? 0 : -1, kXA2ABwd + 0);
  } else if (k == 5 || k == 6) {
    if (split) slab_exchange(FG_PLAN_A2A_BACKWARD, k - 4, kXA2ABwd + k - 4);
  } else if (k >= 7 && k <= 9) {
    if (split) {
      comm_wait(kXA2ABwd + k - 7);
      backward(k - 7, 1);
      if (k == 9) finish();
    } else if (k == 7) {
      comm_wait(kXA2ABwd);
      backward(0, NC);
      finish();
    }
  } else {
    throw std::runtime_error("slab_chain_step: k out of range");
  }
}

// strain-state pipeline (laminate mixing, grids the tiled sweep does not fit, mixed BC with relaxation)
void Solver::slab_front_exact(bool sum_tau) {
  comm_wait(kXHaloU);
  comm_wait(kXSums);
  if (su_valid_ && eps_stale_) slab_materialise_eps();
  if (opt_.mixing == kMixLaminate && !normals_) throw std::runtime_error("laminate mixing needs interface normals");
  FieldPtrs<3> nrm;
  for (int c = 0; c < 3; ++c) nrm.p[c] = normals_ ? normals_ + (long)c * g_.n : nullptr;
  launch_stress(g_, stress_params(opt_.mu_0, opt_.lambda_0, 1.0), ptrs6(eps_), phase_ptrs(), nrm, ptrs6(tau_), derr_, stream_);
  sum_tau = sum_tau || opt_.mode == 2;   // viscosity: the Delta operator needs <tau> in every pass
  if (sum_tau) launch_sum6(g_, ptrs6(tau_), false, partial_, dscal_ + kSlotMean, stream_);
  const long plane = g_.nyzp, last = (long)(g_.nx - 1) * g_.nyzp;
  launch_copy(tau_ + 0 * g_.n + last, halo_[1], plane, stream_);     // tau0, my last plane  -> right (x-1 there)
  launch_copy(tau_ + 5 * g_.n, halo_[0], plane, stream_);            // tau5, my first plane -> left  (x+1 there)
  launch_copy(tau_ + 4 * g_.n, halo_[0] + plane, plane, stream_);    // tau4
  slab_exchange(FG_PLAN_HALO_TAU, 0, kXHaloTau);
  if (sum_tau) slab_reduce(kSlotMean, 6, false);
}

void Solver::slab_div_exact() {
  comm_wait(kXHaloTau);
  const long plane = g_.nyzp;
  XHalo h = {{halo_[2], nullptr}, {halo_[3], halo_[3] + plane}};
  launch_div(g_, ptrs6(tau_), ptrs3(fu_), h, stream_);
}

void Solver::slab_back_exact(const double* E6, const double* R6) {
  comm_wait(kXHaloU);
  double* un = su_[su_cur_ ^ 1];
  if (opt_.mode == 2) {
    // DeltaOperatorStaggered  F:20438-20452 after the Green operator: eta = (E - coef <tau>) + sym grad u + coef tau, coef =
    // 2 alpha / (4 mu0).  <tau> is the all-reduced sum in kSlotMean (mixed BC:
    // the projector term in R6); tau itself is re-evaluated from the strain the pass started from -- its field was the all-to-all buffer
    comm_wait(kXSums);
    Vec6 Ev;
    for (int c = 0; c < 6; ++c) Ev.v[c] = E6[c] + (R6 ? R6[c] : 0.0);   // mixed BC: alpha MQ:<tau> from the host (pass_exact)
    const double alpha = -1.0, m = 1 / (4 * opt_.mu_0);
    launch_eps_delta_recompute(gu_, strided3(un, ucs_), ptrs6(eps_), stress_params(opt_.mu_0, opt_.lambda_0, 1.0), phase_ptrs(),
                               dscal_ + kSlotMean, (double)nglobal_, Ev, 2 * alpha * m, ptrs6(eps_), partial_, dscal_ + kSlotSumSq,
                               stream_);
    slab_reduce(kSlotSumSq, 6, false);
    return;
  }
  const SlabDims d = slab_dims(nxg_, g_.ny, g_.nz, nranks_, rank_);
  const long lo = slab_lo_plane(d) * d.plane, hi = slab_hi_plane(d) * d.plane;
  XHalo h = {{un + 1 * ucs_ + lo, un + 2 * ucs_ + lo}, {un + 0 * ucs_ + hi, nullptr}};
  Vec6 E, R;
  bool add_R = false;
  for (int c = 0; c < 6; ++c) {
    E.v[c] = E6[c];
    R.v[c] = R6 ? R6[c] : 0.0;
    if (R.v[c] != 0.0) add_R = true;
  }
  launch_eps_norm(g_, strided3(un, ucs_), ptrs6(eps_), E, R, add_R, partial_, dscal_ + kSlotSumSq, h, stream_);
  slab_reduce(kSlotSumSq, 6, false);
}

void Solver::slab_materialise_eps() {
  comm_wait(kXHaloU);
  double* u = su_[su_cur_];
  const SlabDims d = slab_dims(nxg_, g_.ny, g_.nz, nranks_, rank_);
  const long lo = slab_lo_plane(d) * d.plane, hi = slab_hi_plane(d) * d.plane;
  XHalo h = {{u + 1 * ucs_ + lo, u + 2 * ucs_ + lo}, {u + 0 * ucs_ + hi, nullptr}};
  Vec6 E, R;
  for (int c = 0; c < 6; ++c) E.v[c] = E_cur_[c], R.v[c] = 0.0;
  if (opt_.mode == 1) launch_sc_grad(gu_, u, ptrs3(eps_), E, partial_, dscal_ + kSlotScratch, stream_);   // g = E + grad+ T
  else launch_eps_norm(g_, strided3(u, ucs_), ptrs6(eps_), E, R, false, partial_, dscal_ + kSlotScratch, h, stream_);
  eps_stale_ = false;
}

void Solver::slab_adopt(const double* E6, bool u_is_state) {
  su_cur_ ^= 1;
  su_valid_ = u_is_state;
  for (int c = 0; c < 6; ++c) E_cur_[c] = E6[c];
}

void Solver::slab_reset_state() {
  solve_time_ = 0.0;
  residuals_.clear();
  iterations_ = 0;
  cancel_ = false;
  u_valid_ = false;
  for (int i = 0; i < 6; ++i) F00_[i] = 0.0;
}

// ===================================================================================================== group side
void SlabGroup::check_members() const {
  if (m_.empty()) throw std::runtime_error("empty slab group");
  const Solver& a = *m_[0];
  if (a.opt_.gamma_scheme != 0) throw std::runtime_error("slab-decomposed solvers run the staggered Green operator");
  if (a.opt_.mode == 2 && (a.opt_.mixing != kMixVoigt || a.opt_.bc_relax != 1.0))
    throw std::runtime_error("viscosity mode supports Voigt mixing and bc_relax = 1 only");
  if (a.pt_.n < 1) throw std::runtime_error("No materials specified");
  for (Solver* s : m_) {
    if (s->nranks_ > 1 && !s->comm_) throw std::runtime_error("slab solver is not connected to a transport (fg_slab_connect_*)");
    if (s->opt_.mixing != a.opt_.mixing || s->pt_.n != a.pt_.n || s->opt_.u_loop != a.opt_.u_loop || s->opt_.u_tile != a.opt_.u_tile ||
        s->opt_.slab_split != a.opt_.slab_split || s->opt_.slab_interleave != a.opt_.slab_interleave || s->opt_.method != a.opt_.method ||
        s->opt_.mode != a.opt_.mode)
      throw std::runtime_error("the members of a slab group must carry the same options and materials");
  }
}

// heat / porous on slabs: the potential-based fast path only (prescribed mean gradients, a grid the tiled sweep fits)
void SlabGroup::require_scalar_fast(bool allow_mixed_bc) const {
  if (m_[0]->opt_.mode == 1 && !fast_ok(allow_mixed_bc))
    throw std::runtime_error("heat / porous on slab-decomposed solvers: Voigt mixing, u_loop=2, bc_relax=1 (method=cg: prescribed "
                             "mean gradients)");
}

void SlabGroup::synchronize() {
  for (Solver* s : m_) {
    FG_HIP_CHECK(hipSetDevice(s->device_));
    FG_HIP_CHECK(hipStreamSynchronize(s->stream_));
    if (s->comm_stream_ != s->stream_) FG_HIP_CHECK(hipStreamSynchronize(s->comm_stream_));
  }
}

bool SlabGroup::fast_ok(bool allow_mixed_bc) const {
  for (Solver* s : m_)
    if (!s->slab_fast_ok(allow_mixed_bc)) return false;
  return true;
}

void SlabGroup::prepare() {
  for (Solver* s : m_) {
    FG_HIP_CHECK(hipSetDevice(s->device_));
    s->slab_alloc();
  }
  if (fast_ok(true))
    for (Solver* s : m_) {
      s->slab_moduli_step();
      if (s->opt_.mixing == kMixLaminate) s->build_laminate_lists();   // allocates and synchronises: not inside a pass
    }
}

// Waits for the sums (and the flag word) of the last reduction.  The error decision is taken on the REDUCED flag: when a
// kernel of any rank raised its error flag (a laminate voxel with more than two phases that only one slab contains, say),
// every rank throws here, in the same pass.
void SlabGroup::wait_norms() {
  for (Solver* s : m_) FG_HIP_CHECK(hipEventSynchronize(s->ev_norm_));
  bool any = false, mine = false;
  for (Solver* s : m_) {
    any = any || s->hscal_[kSlotFlag] != 0.0;
    mine = mine || *s->herr_ != 0;
  }
  if (!any) return;
  for (Solver* s : m_) FG_HIP_CHECK(hipMemsetAsync(s->derr_, 0, sizeof(int), s->stream_));
  const char* where = mine ? "(stress)" : "(stress, on another rank)";
  if (m_[0]->opt_.mixing == kMixLaminate)
    throw std::runtime_error(std::string("The laminate mixing rule supports only two phase mixtures ") + where);
  throw std::runtime_error(std::string("device kernel reported an error ") + where);
}

bool SlabGroup::stop_requested() const { return m_[0]->hscal_[kSlotFlag + 1] != 0.0; }

// all members have left their local contribution in dscal_[slot..slot+n): reduce over ranks, bring to every host
void SlabGroup::reduce_and_fetch(int slot, int n, bool min_op) {
  for (Solver* s : m_) s->slab_reduce(slot, n, min_op);
  for (Solver* s : m_) {
    FG_HIP_CHECK(hipMemcpyAsync(s->hscal_ + slot, s->dscal_ + slot, n * sizeof(double), hipMemcpyDeviceToHost, s->comm_stream_));
    FG_HIP_CHECK(hipMemcpyAsync(s->hscal_ + kSlotFlag, s->dscal_ + kSlotFlag, 2 * sizeof(double), hipMemcpyDeviceToHost,
                                s->comm_stream_));
    FG_HIP_CHECK(hipMemcpyAsync(s->herr_, s->derr_, sizeof(int), hipMemcpyDeviceToHost, s->comm_stream_));
    FG_HIP_CHECK(hipEventRecord(s->ev_norm_, s->comm_stream_));
  }
  wait_norms();
  for (Solver* s : m_) s->comm_wait(kXSums);   // later writes of dscal_ on the compute stream follow the reduction
}

bool SlabGroup::agree_on_voting() {
  if (m_[0]->nranks_ == 1) return false;
  double v[2] = {0.0, 0.0};
  for (Solver* s : m_)
    if (s->cb_) v[0] = 1.0;
  vote(v);
  return v[0] != 0.0;
}

// Sum over all ranks of two small host values (votes: a callback asked to stop, a rank was cancelled).
void SlabGroup::vote(double* v2) {
  for (Solver* s : m_) {
    s->hscal_[kSlotMisc] = s == m_[0] ? v2[0] : 0.0;
    s->hscal_[kSlotMisc + 1] = s == m_[0] ? v2[1] : 0.0;
    FG_HIP_CHECK(hipMemcpyAsync(s->dscal_ + kSlotMisc, s->hscal_ + kSlotMisc, 2 * sizeof(double), hipMemcpyHostToDevice, s->stream_));
  }
  reduce_and_fetch(kSlotMisc, 2, false);
  v2[0] = m_[0]->hscal_[kSlotMisc];
  v2[1] = m_[0]->hscal_[kSlotMisc + 1];
}

void SlabGroup::mean_strain(double* out6) {
  check_members();
  prepare();
  for (Solver* s : m_) {
    if (s->su_valid_ && s->eps_stale_) s->slab_materialise_eps();
    launch_sum6(s->g_, s->ptrs6(s->eps_), false, s->partial_, s->dscal_ + kSlotMean, s->stream_);
  }
  reduce_and_fetch(kSlotMean, 6, false);
  for (int c = 0; c < 6; ++c) out6[c] = m_[0]->hscal_[kSlotMean + c] / (double)m_[0]->nglobal_;
}

void SlabGroup::mean_stress(double* out6) {
  check_members();
  prepare();
  for (Solver* s : m_) {
    if (s->su_valid_ && s->eps_stale_) s->slab_materialise_eps();
    FieldPtrs<3> nrm;
    for (int c = 0; c < 3; ++c) nrm.p[c] = s->normals_ ? s->normals_ + (long)c * s->g_.n : nullptr;
    // meanPK1: alpha /= nxyz (global), accumulate  F:12318-12340
    if (s->opt_.mode == 1)
      launch_sc_flux_mean(s->g_, s->scalar_params(0.0, 1.0 / (double)s->nglobal_), s->ptrs3(s->eps_), s->phase_ptrs(), s->partial_,
                          s->dscal_ + kSlotMean, s->stream_);
    else
      launch_stress_mean(s->g_, s->stress_params(0.0, 0.0, 1.0 / (double)s->nglobal_), s->ptrs6(s->eps_), s->phase_ptrs(), nrm,
                         s->partial_, s->dscal_ + kSlotMean, s->derr_, s->stream_);
  }
  reduce_and_fetch(kSlotMean, 6, false);
  for (int c = 0; c < 6; ++c) out6[c] = m_[0]->hscal_[kSlotMean + c];
}

// meanW  F:12239-12262 over all slabs
double SlabGroup::mean_energy() {
  check_members();
  prepare();
  for (Solver* s : m_) {
    if (s->opt_.mode == 1) throw std::runtime_error("the energy error estimator is not available in heat / porous mode");
    if (s->su_valid_ && s->eps_stale_) s->slab_materialise_eps();
    FieldPtrs<3> nrm;
    for (int c = 0; c < 3; ++c) nrm.p[c] = s->normals_ ? s->normals_ + (long)c * s->g_.n : nullptr;
    launch_energy_mean(s->g_, s->stress_params(0.0, 0.0, 1.0), s->ptrs6(s->eps_), s->phase_ptrs(), nrm, s->partial_,
                       s->dscal_ + kSlotMean, s->derr_, s->stream_);
  }
  reduce_and_fetch(kSlotMean, 6, false);
  return m_[0]->hscal_[kSlotMean] / (double)m_[0]->nglobal_;
}

// the estimators that measure a mean of the strain field (Solver::estimator_begin / estimator_update), collectively: every
// rank holds the same all-reduced means and so the same estimator state
void SlabGroup::estimator_begin(bool fresh) {
  Solver& a = *m_[0];
  if (a.opt_.error_estimator < 2) return;
  if (a.opt_.mode == 1) throw std::runtime_error("heat / porous mode supports the error estimators epsilon and residual");
  double m[6] = {0, 0, 0, 0, 0, 0};
  double w = 0.0;
  if (a.opt_.error_estimator == 2 && !fresh) mean_stress(m);
  if (a.opt_.error_estimator == 3 && !fresh) w = mean_energy();
  for (Solver* s : m_) {
    if (a.opt_.error_estimator == 2) s->est_.start_sigma(m);
    if (a.opt_.error_estimator == 3) s->est_.start_energy(w);
  }
}

void SlabGroup::estimator_update(double* abs_err, double* rel_err) {
  Solver& a = *m_[0];
  if (a.opt_.error_estimator == 2) {
    double m[6];
    mean_stress(m);
    double ae = 0.0, re = 0.0;
    for (Solver* s : m_) s->est_.update_sigma(m, &ae, &re);
    *abs_err = ae, *rel_err = re;
  } else if (a.opt_.error_estimator == 3) {
    const double w = mean_energy();
    double ae = 0.0, re = 0.0;
    for (Solver* s : m_) s->est_.update_energy(w, &ae, &re);
    *abs_err = ae, *rel_err = re;
  } else if (a.opt_.error_estimator == 4) {
    *abs_err = *rel_err = 1.0;
  }
}

double SlabGroup::volume_fraction(int p) {
  check_members();
  prepare();
  for (Solver* s : m_) {
    if (p < 0 || p >= s->pt_.n) throw std::runtime_error("phase index out of range");
    launch_sum1(s->g_, s->phi_ + (long)p * s->g_.n, s->partial_, s->dscal_ + kSlotMisc, s->stream_);
  }
  reduce_and_fetch(kSlotMisc, 1, false);
  return m_[0]->hscal_[kSlotMisc] / (double)m_[0]->nglobal_;
}

// calcRefMaterial  F:22283-22313 with the extreme tangent eigenvalues taken over all slabs
void SlabGroup::calc_ref_material() {
  check_members();
  prepare();
  for (Solver* s : m_) {
    if (s->opt_.mode == 1)
      launch_sc_minmax(s->g_, s->scalar_params(0.0, 1.0), s->phase_ptrs(), s->partial_, s->dscal_ + kSlotMinMax, s->stream_);
    else
      launch_tangent_minmax(s->g_, s->phase_table(), s->opt_.mixing, s->phase_ptrs(), s->partial_, s->dscal_ + kSlotMinMax, s->derr_,
                            s->stream_);
  }
  reduce_and_fetch(kSlotMinMax, 2, true);   // stored as (min, -max): one element-wise minimum serves both
  double lambda_min = m_[0]->hscal_[kSlotMinMax], lambda_max = -m_[0]->hscal_[kSlotMinMax + 1];
  if (lambda_min < 0) lambda_min = 0;   // F:12183-12223
  double mu_0 = 0.5 * (lambda_min + lambda_max);
  mu_0 *= 0.5 * m_[0]->opt_.ref_scale;
  for (Solver* s : m_) {
    s->opt_.mu_0 = mu_0;
    s->recompute_bc();   // F:22312
  }
}

// bc_error  F:21129-21161
double SlabGroup::bc_error(const double* E_cur, const double* S_cur) {
  Solver& a = *m_[0];
  double Emean[6], Smean[6], PE[6], QS[6], PEc[6], d[6];
  mean_strain(Emean);
  mean_stress(Smean);
  voigt_mv(a.BC_P_, Emean, PE);
  voigt_mv(a.BC_Q_, Smean, QS);
  voigt_mv(a.BC_P_, E_cur, PEc);
  const double norm_E = voigt_norm2(PEc);
  for (int i = 0; i < 6; ++i) d[i] = PE[i] - E_cur[i];
  const double err_F = voigt_norm2(d) / ((norm_E < a.opt_.bc_tol) ? 1 : norm_E);
  const double norm_S = voigt_norm2(S_cur);
  for (int i = 0; i < 6; ++i) d[i] = QS[i] - S_cur[i];
  const double err_S = voigt_norm2(d) / ((norm_S < a.opt_.bc_tol) ? 1 : norm_S);
  return err_F > err_S ? err_F : err_S;
}

// one displacement pass on every member: the sweep (norms of eps_k), then -- speculatively, the host has not seen the
// norms yet -- the whole transform chain to u_{k+1}; adopted by the caller if the loop goes on.  chain = false leaves the
// chain to a later pass_fast_chain() (a run with convergence callbacks decides first, see SlabGroup::run)
void SlabGroup::pass_fast(const double* E_cur, bool sum_tau, bool chain) {
  for (Solver* s : m_) s->slab_front_fast(E_cur, sum_tau);
  for (Solver* s : m_) s->slab_front_laminate(sum_tau);
  for (Solver* s : m_) s->slab_fetch_norms(sum_tau ? 12 : 6);
  if (chain) pass_fast_chain();
}

void SlabGroup::pass_fast_chain() {
  for (int k = 1; k <= 9; ++k)
    for (Solver* s : m_) s->slab_chain_step(k);
}

// one pass of the strain-state pipeline: eps_ -> eps_ (and the displacement it was built from in su_[next]); adopts
void SlabGroup::pass_exact(const double* E6, bool mixed_bc) {
  Solver& a = *m_[0];
  const double alpha = -1.0;
  double F00[6] = {0, 0, 0, 0, 0, 0};
  if (a.opt_.bc_relax != 1.0) mean_strain(F00);   // F:20563-20565: mean of the operator's argument
  for (Solver* s : m_) s->slab_front_exact(mixed_bc);
  double F0[6] = {0, 0, 0, 0, 0, 0};
  if (mixed_bc) {   // initBCProjector  F:20228-20239: <tau> over all slabs
    for (Solver* s : m_) {
      FG_HIP_CHECK(hipMemcpyAsync(s->hscal_ + kSlotMean, s->dscal_ + kSlotMean, 6 * sizeof(double), hipMemcpyDeviceToHost,
                                  s->comm_stream_));
      FG_HIP_CHECK(hipEventRecord(s->ev_norm_, s->comm_stream_));
    }
  }
  for (Solver* s : m_) s->slab_div_exact();
  for (int k = 1; k <= 9; ++k)
    for (Solver* s : m_) s->slab_chain_step(k);
  if (mixed_bc) {
    for (Solver* s : m_) FG_HIP_CHECK(hipEventSynchronize(s->ev_norm_));
    for (int c = 0; c < 6; ++c) F0[c] = a.hscal_[kSlotMean + c] / (double)a.nglobal_;
  }
  // applyBCProjector  F:20247-20270: R = alpha*(bc_relax*MQ:F0 - (1-bc_relax)*M:(QC0:F00))
  double R[6], t1[6], t2[6], t3[6];
  voigt_mv(a.BC_MQ_, F0, t1);
  voigt_mv(a.BC_QC0_, F00, t2);
  voigt_mv(a.BC_M_, t2, t3);
  bool add_R = false;
  for (int c = 0; c < 6; ++c) {
    R[c] = (!mixed_bc && a.opt_.bc_relax == 1.0) ? 0.0 : alpha * (a.opt_.bc_relax * t1[c] - (1 - a.opt_.bc_relax) * t3[c]);
    if (R[c] != 0.0) add_R = true;
  }
  for (Solver* s : m_) s->slab_back_exact(E6, R);
  for (Solver* s : m_) s->slab_fetch_norms(6);
  for (Solver* s : m_) {
    s->slab_adopt(E6, !add_R && a.opt_.mode == 0);   // viscosity: eta is not E + sym grad u, the strain field is the state
    s->eps_stale_ = false;
  }
}

void SlabGroup::iterate(const double* E6, int n) {
  check_members();
  prepare();
  require_scalar_fast(false);
  Solver& a = *m_[0];
  if (a.opt_.mode == 1)
    for (Solver* s : m_)
      if (!s->su_valid_) {   // no potential yet: T = 0 (g = E), like the start of a run
        s->comm_wait(kXHaloU);
        FG_HIP_CHECK(hipMemsetAsync(s->su_[s->su_cur_], 0, 3 * (size_t)s->ucs_ * sizeof(double), s->stream_));
        s->su_valid_ = true;
        s->eps_stale_ = true;
        for (int i = 0; i < 6; ++i) s->E_cur_[i] = E6[i];
      }
  const bool mixed_bc = !(frobenius(a.BC_MQ_) < kEps);
  const bool fast = fast_ok(false);
  int i = 0;
  if (fast) {
    bool have_u = true;
    for (Solver* s : m_) have_u = have_u && s->su_valid_;
    if (!have_u && n > 0) {   // no displacement yet: one strain-state pass leaves u (with its halo planes) and eps
      pass_exact(E6, false);
      ++i;
    }
    for (; i < n; ++i) {
      pass_fast(m_[0]->E_cur_, false, true);
      for (Solver* s : m_) {
        s->slab_adopt(E6, true);
        s->eps_stale_ = true;
      }
    }
    return;
  }
  for (; i < n; ++i) pass_exact(E6, mixed_bc);
}

// LSSolver::run F:21247-21398 -> runLoadsteppingSolver (one load step) -> runBasic F:21716-21805, stop rule of
// _converged F:21177-21244 -- the collective counterpart of Solver::run
// runLoadsteppingSolver  F:21584-21685 on the slabs: step i prescribes params[i] * (E, S) and continues from the field of step
// i - 1 (no extrapolation); LSSolver::run is the one-step case.
bool SlabGroup::run(const double* E6, const double* S6) {
  const double one = 1.0;
  return run_load_steps(E6, S6, &one, 1, 0, nullptr, nullptr);
}

bool SlabGroup::run_load_steps(const double* E6, const double* S6, const double* params, int nparams, int first,
                               LoadstepCallback step_cb, void* user) {
  check_members();
  Solver& a = *m_[0];
  if (nparams < 1 || first < 0 || !params) throw std::runtime_error("invalid load steps");
  if (a.opt_.loadstep_extrapolation_order > 0 && nparams - first > 1)
    throw std::runtime_error("load-step extrapolation is not available on slab-decomposed solvers");
  double Emax[6], Smax[6];
  for (int i = 0; i < 6; ++i) {
    Emax[i] = E6[i];
    Smax[i] = S6 ? S6[i] : 0.0;
  }
  for (Solver* s : m_) {
    FG_HIP_CHECK(hipSetDevice(s->device_));
    s->slab_reset_state();
    s->recompute_bc();   // F:21354
  }
  {
    const double se = std::sqrt(kEps);
    double t[6];
    voigt_mv(a.BC_P_, Smax, t);
    if (norm2(t, 6) > se * norm2(Smax, 6)) throw std::runtime_error("Incompatible stress boundary condition specified");
    voigt_mv(a.BC_Q_, Emax, t);
    if (norm2(t, 6) > se * norm2(Emax, 6)) throw std::runtime_error("Incompatible strain boundary condition specified");
  }
  for (int istep = first; istep < nparams; ++istep) {
    double E[6], S[6];
    for (int i = 0; i < 6; ++i) E[i] = params[istep] * Emax[i], S[i] = params[istep] * Smax[i];
    const bool fresh = istep == first;
    const bool failed = a.opt_.method == 1 ? run_cg(E, S, fresh) : run_step(E, S, fresh);
    // The step's outcome and the load-step callback's answer are agreed over the ranks: a callback that exists on one
    // rank only (or answers differently per rank) must not let that rank return while the others enter the next step's
    // collectives.  One tiny all-reduce per load step.
    double v[2] = {failed ? 1.0 : 0.0, 0.0};
    if (!failed && step_cb && step_cb(user, istep)) v[1] = 1.0;   // "Loadstep callback break request."
    if (a.nranks_ > 1) vote(v);
    if (v[0] != 0.0 || v[1] != 0.0) return true;
  }
  return false;
}

// norm of the strain field a continuing step finds (EpsilonErrorEstimator constructor  F:14612-14618), all ranks
double SlabGroup::current_norm9() {
  prepare();
  for (Solver* s : m_) {
    if (s->su_valid_ && s->eps_stale_) s->slab_materialise_eps();
    launch_sum6(s->g_, s->ptrs6(s->eps_), true, s->partial_, s->dscal_ + kSlotMean, s->stream_);
  }
  reduce_and_fetch(kSlotMean, 6, false);
  double s9 = 0.0;
  for (int c = 0; c < 6; ++c) {
    const double m = std::sqrt(m_[0]->hscal_[kSlotMean + c] / (double)m_[0]->nglobal_);
    s9 += m * m * ((c >= 3) ? 2.0 : 1.0);
  }
  return std::sqrt(s9);
}

// runSolver / runBasic  F:21400-21433, F:21716-21805 for one load step, stop rule of _converged F:21177-21244 -- the
// collective counterpart of Solver::run_one_step
bool SlabGroup::run_step(const double* E0, const double* S0, bool fresh) {
  Solver& a = *m_[0];
  const double t_start = now_seconds();
  prepare();
  require_scalar_fast(true);
  const bool fast_allowed = fast_ok(true);
  bool fast = fast_allowed;
  double prev = 0.0;   // EpsilonErrorEstimator  F:14591-14637: norms of the field the step starts from
  if (fresh) {
    for (Solver* s : m_) {
      FG_HIP_CHECK(hipMemsetAsync(s->eps_, 0, 6 * (size_t)s->g_.n * sizeof(double), s->stream_));   // F:21379
      s->comm_wait(kXHaloU);
      FG_HIP_CHECK(hipMemsetAsync(s->su_[s->su_cur_], 0, 3 * (size_t)s->ucs_ * sizeof(double), s->stream_));   // u_1 = 0 (eps_1 = E)
      s->su_valid_ = fast;
      s->eps_stale_ = fast;
      for (int i = 0; i < 6; ++i) s->E_cur_[i] = E0[i];
    }
  } else {
    prev = current_norm9();
  }
  estimator_begin(fresh);
  for (Solver* s : m_) s->in_run_ = true;
  // a continuing step in the displacement loop: the state is u of the previous load (eps = E_old + sym grad u); one
  // unrecorded pass turns it into u' with eps' = E_new + sym grad u', the field the reference's first iteration of the step
  // produces (Solver::run_one_step)
  bool carry = !fresh && fast;
  for (Solver* s : m_) carry = carry && s->su_valid_;

  // Stop decisions are collective.  Callbacks may be installed on some ranks only (rank 0 printing its progress) and may
  // answer differently; whether any rank has one is agreed once per run, and if so every pass ends with a vote on the
  // callbacks' answers before the transform chain of the next pass is enqueued (no speculation: a vote behind the
  // speculative exchanges would wait for them).  Without callbacks only fg_cancel can ask for a stop from outside; it
  // travels with the flag word of the next pass's reduction.
  const bool voting = agree_on_voting();

  long iter = 1;
  bool update_ref = a.opt_.update_ref != 0;
  double E[6], E_next[6];
  for (int i = 0; i < 6; ++i) E[i] = E0[i];
  bool failed = false;
  const double small = std::numeric_limits<double>::min();
  const double nglobal = (double)a.nglobal_;

  for (;;) {
    if (update_ref) {
      calc_ref_material();
      double t1[6], t2[6], t3[6];   // calcBCMean  F:20242-20245
      voigt_mv(a.BC_QC0_, E0, t1);
      for (int i = 0; i < 6; ++i) t2[i] = S0[i] - t1[i];
      voigt_mv(a.BC_M_, t2, t3);
      for (int i = 0; i < 6; ++i) E[i] = E0[i] + a.opt_.bc_relax * t3[i];
      update_ref = false;
      if (fast) prepare();   // effective moduli are independent of the reference material, but may not exist yet
    }
    const bool mixed_bc = !(frobenius(a.BC_MQ_) < kEps);
    if (carry) {
      carry = false;
      if (mixed_bc) {
        fast = false;   // the correction of the prescribed mean needs <tau> of the old field: strain-state passes for this step
      } else {
        pass_fast(a.E_cur_, false, true);
        for (Solver* s : m_) {
          s->slab_adopt(E, true);
          s->eps_stale_ = true;
        }
      }
    }
    bool all_u = true;
    for (Solver* s : m_) all_u = all_u && s->su_valid_;
    bool pending = false;
    if (fast && all_u) {
      if (iter == 1 && fresh)
        for (Solver* s : m_)
          for (int i = 0; i < 6; ++i) s->E_cur_[i] = E[i];   // eps_1 = E (u_1 = 0)
      pass_fast(a.E_cur_, mixed_bc, !voting);
      pending = true;
    } else {
      pass_exact(E, mixed_bc);
      // the pass leaves a displacement unless the projector's correction went into the strain: go on in displacement space
      // after the unrecorded pass of a continuing state (carry)
      bool have_u = fast && !mixed_bc;
      for (Solver* s : m_) have_u = have_u && s->su_valid_;
      if (have_u) carry = true;
      else fast = false;
    }
    wait_norms();
    for (int i = 0; i < 6; ++i) E_next[i] = E[i];
    if (pending && mixed_bc) {
      // applyBCProjector  F:20247-20270 with bc_relax = 1: eps_{k+1} = E + alpha MQ:<tau_k> + sym grad u_{k+1}
      double F0[6], t1[6];
      for (int c = 0; c < 6; ++c) F0[c] = a.hscal_[kSlotMean + c] / nglobal;
      voigt_mv(a.BC_MQ_, F0, t1);
      for (int c = 0; c < 6; ++c) E_next[c] = E[c] - t1[c];   // alpha = -1  (F:20575)
    }

    // component_norm + fix_dim + norm_2 over 9 mirrored entries  F:10127-10138, F:14600-14609, F:14627
    double mm[6], s9 = 0.0;
    for (int c = 0; c < 6; ++c) {
      const double ss = a.hscal_[kSlotSumSq + c];
      for (Solver* s : m_) s->sumsq_[c] = ss;
      mm[c] = std::sqrt(ss / nglobal);
    }
    for (int c = 0; c < 6; ++c) s9 += mm[c] * mm[c];
    for (int c = 3; c < 6; ++c) s9 += mm[c] * mm[c];
    const double cur = std::sqrt(s9);
    double abs_err = std::fabs(prev - cur);
    double rel_err = abs_err / (small + cur);
    prev = cur;
    if (a.opt_.error_estimator >= 2) estimator_update(&abs_err, &rel_err);   // sigma / energy / none: F:14410-14587

    // _converged  F:21177-21244.  rel_err comes from all-reduced sums and is the same on every rank; so is the stop word.
    if (std::isnan(rel_err) || stop_requested()) {
      failed = true;
      break;
    }
    for (Solver* s : m_) s->residuals_.push_back(rel_err);
    bool stop = false, cancelled = false;
    for (Solver* s : m_) {
      if (s->cb_ && s->cb_(s->cb_user_)) stop = true;
      if (s->cancel_) cancelled = true;   // cancelled from inside the callback
    }
    if (voting) {
      double v[2] = {stop ? 1.0 : 0.0, cancelled ? 1.0 : 0.0};
      vote(v);
      stop = v[0] != 0.0;
      cancelled = v[1] != 0.0;
    }
    if (a.nranks_ > 1 && !voting) cancelled = false;   // an asynchronous fg_cancel: acted upon through the next flag word
    if (stop) break;
    if (cancelled) {
      failed = true;
      break;
    }
    if (iter >= a.opt_.maxiter) break;
    if (rel_err <= a.opt_.tol || abs_err <= a.opt_.abs_tol) {
      if (bc_error(E0, S0) <= a.opt_.bc_tol) break;
    }
    if (pending) {
      if (voting) pass_fast_chain();
      for (Solver* s : m_) {
        s->slab_adopt(E_next, true);
        s->eps_stale_ = true;
      }
    }
    iter++;
  }
  for (Solver* s : m_) {
    s->in_run_ = false;
    s->iterations_ = iter;
    if (s->su_valid_ && s->eps_stale_) s->slab_materialise_eps();
  }
  synchronize();
  const double dt = now_seconds() - t_start;
  for (Solver* s : m_) s->solve_time_ += dt;
  return failed;
}

// runCGElasticity  F:23153-23247 on the slabs, carried in displacement space like Solver::run_cg_u (eps = E + grad_s u_e;
// r, p, w = grad_s u_r, u_p, u_w).  u_e = su_[cur], u_w = su_[cur ^ 1] (where the transform chain leaves its result, halo
// planes exchanged), u_r / u_p = scg_.  The vector updates are point-wise and linear with coefficients every rank forms
// from the same all-reduced sums, so applying them to the spare planes as well keeps the halo planes of u_e, u_r and u_p
// valid without any further exchange: per iteration the only traffic beyond the operator's own (two all-to-alls, halo of
// u_w) are two all-reduces, p:(p - w) and the seven sums of the stop rule (norms of eps, r:r) -- innerProductL2
// F:20955-21038 with the partial sums of the slabs added in rank order by the transport.
bool SlabGroup::run_cg(const double* E6, const double* S6, bool fresh) {
  Solver& a = *m_[0];
  double E0[6], S0[6];
  for (int i = 0; i < 6; ++i) {
    E0[i] = E6[i];
    S0[i] = S6 ? S6[i] : 0.0;
  }
  prepare();
  // CG restarts every step from eps = E (F:23184); only the estimator remembers the field the step found (F:14612-14618)
  const double prev0 = fresh ? 0.0 : current_norm9();
  estimator_begin(fresh);
  if (a.opt_.mode == 1) {
    if (norm2(S0, 6) != 0.0 || !(frobenius(a.BC_Q_) < kEps))
      throw std::runtime_error("method=cg in heat / porous mode on slab-decomposed solvers: prescribed mean gradients");
    require_scalar_fast(false);
    return run_cg_scalar(E0, prev0);
  }
  // mixed boundary conditions, grids the tiled sweep does not fit, u_loop < 2: the strain-space form (the vectors of
  // runCGElasticity as 6-component fields, the operator = one pass of the strain-state pipeline)
  // (and the estimators that measure a mean of the strain field: the iterate is a stored field there)
  if (norm2(S0, 6) != 0.0 || !(frobenius(a.BC_Q_) < kEps) || !fast_ok(false) || a.opt_.error_estimator >= 2)
    return run_cg_strain(E0, S0, prev0);
  const double t_start = now_seconds();
  if (a.opt_.update_ref) {
    calc_ref_material();
    prepare();
  }
  const bool voting = agree_on_voting();
  const bool residual_est = a.opt_.error_estimator == 1;
  const double small = std::numeric_limits<double>::min();
  const double nglobal = (double)a.nglobal_;
  const int blk[2] = {kSlotCg, kSlotCg + 8}, s0 = kSlotCg + 16;
  Vec6 E, Z;
  for (int i = 0; i < 6; ++i) E.v[i] = E0[i], Z.v[i] = 0.0;

  auto u_e = [](Solver* s) { return s->su_[s->su_cur_]; };
  auto u_w = [](Solver* s) { return s->su_[s->su_cur_ ^ 1]; };
  auto u_r = [](Solver* s) { return s->cgs_r_; };
  auto u_p = [](Solver* s) { return s->cgs_p_; };
  // Fused form (option cg_fused, as Solver::run_cg_u): p:(p - w) and the update of eps, r with their norms as two tiled sweeps
  // (the own planes of the alternate buffers; their spare planes by a point-wise kernel), the direction update inside the
  // operator's sweep (Voigt mixing).
  bool fused = a.opt_.cg_fused != 0;
  // u_w = operator(u_in) with prescribed mean Eadd: sweep + transform chain; the halo planes of u_w are on their way
  auto apply = [&](bool from_p, const double* Eadd) {
    for (Solver* s : m_) s->slab_front_fast(Eadd, false, from_p ? u_p(s) : u_e(s), false);
    for (Solver* s : m_) s->slab_front_laminate(false, false);
    pass_fast_chain();
  };
  auto fetch7 = [&](int slot) {
    for (Solver* s : m_) {
      FG_HIP_CHECK(hipMemcpyAsync(s->hscal_ + kSlotCg, s->dscal_ + slot, 7 * sizeof(double), hipMemcpyDeviceToHost, s->comm_stream_));
      FG_HIP_CHECK(hipMemcpyAsync(s->hscal_ + kSlotFlag, s->dscal_ + kSlotFlag, 2 * sizeof(double), hipMemcpyDeviceToHost,
                                  s->comm_stream_));
      FG_HIP_CHECK(hipMemcpyAsync(s->herr_, s->derr_, sizeof(int), hipMemcpyDeviceToHost, s->comm_stream_));
      FG_HIP_CHECK(hipEventRecord(s->ev_norm_, s->comm_stream_));
    }
  };
  auto direction_update = [&](int cur, int nxt) {   // p = r + beta p, beta = delta / gamma
    for (Solver* s : m_) {
      s->comm_wait(kXSums);
      launch_cgu_axpy(1, s->gu_, strided3(u_e(s), s->ucs_), strided3(u_p(s), s->ucs_), strided3(u_r(s), s->ucs_),
                      strided3(u_w(s), s->ucs_), s->dscal_, blk[nxt] + 6, blk[cur] + 6, nglobal, small, s->stream_, s->ucs_);
    }
  };

  for (Solver* s : m_) {
    s->slab_cg_alloc();
    s->cgs_r_ = s->scg_;
    s->cgs_p_ = s->scg_ + 3 * s->ucs_;
  }
  {
    // every member (every rank) must take the same form: the decision is the logical AND over the group -- of the option
    // as well as of the allocation (a rank whose cg_fused says 0 votes too: the vote is unconditional, so no rank can miss
    // the all-reduce the others enter)
    double v[2] = {fused ? 0.0 : 1.0, 0.0};
    if (fused)
      for (Solver* s : m_)
        if (!s->slab_cg_alloc_fused()) v[0] = 1.0;
    if (a.nranks_ > 1) vote(v);
    fused = v[0] == 0.0;
  }
  const bool fused_dir = fused && a.opt_.mixing == kMixVoigt;
  for (Solver* s : m_) {
    if (fused) {
      s->cgs_ra_ = s->scg2_;
      s->cgs_pa_ = s->scg2_ + 3 * s->ucs_;
    }
    s->comm_wait(kXHaloU);
    FG_HIP_CHECK(hipMemsetAsync(u_e(s), 0, 3 * (size_t)s->ucs_ * sizeof(double), s->stream_));   // eps_0 = E
    s->su_valid_ = true;
    s->eps_stale_ = true;
    s->in_run_ = true;
    for (int i = 0; i < 6; ++i) s->E_cur_[i] = E0[i];
  }
  auto apply_dir = [&](int cur, int nxt) {   // fused: p = r + beta p inside the sweep; u_w = operator(u_p)
    for (Solver* s : m_) {
      s->comm_wait(kXSums);
      s->slab_front_fast_cg(Z.v, blk[nxt] + 6, blk[cur] + 6, nglobal, small);
    }
    pass_fast_chain();
  };
  apply(false, E.v);   // r = -Gamma0 (C - C0) E  (+ E - eps_0 = 0, adjustResidual F:10012-10022)
  for (Solver* s : m_) {
    s->comm_wait(kXHaloU);
    const size_t f3 = 3 * (size_t)s->ucs_ * sizeof(double);
    FG_HIP_CHECK(hipMemcpyAsync(u_r(s), u_w(s), f3, hipMemcpyDeviceToDevice, s->stream_));
    FG_HIP_CHECK(hipMemcpyAsync(u_p(s), u_w(s), f3, hipMemcpyDeviceToDevice, s->stream_));   // p = r
    launch_cgu_dot(1, s->gu_, strided3(u_e(s), s->ucs_), strided3(u_r(s), s->ucs_), E, s->partial_, s->dscal_ + blk[0], s->stream_);
    s->slab_reduce(blk[0], 7, false);   // gamma_0 = r:r / N + tiny
  }
  double gamma_cur = 0.0, gamma_0 = 0.0;
  if (residual_est) {
    fetch7(blk[0]);
    wait_norms();
    gamma_cur = gamma_0 = a.hscal_[kSlotCg + 6] / nglobal + small;
  }
  double prev = prev0;   // estimator constructed on the field the step starts from
  long iter = 0;
  bool failed = false, applied = false;
  int dir_cur = 0, dir_nxt = 1;   // fused: slots of the pending direction update
  for (;;) {
    const int cur = (int)(iter & 1), nxt = cur ^ 1;
    if (!applied) {
      if (fused_dir && iter > 0) apply_dir(dir_cur, dir_nxt);
      else apply(true, Z.v);   // u_w = operator(u_p)
    }
    applied = false;
    for (Solver* s : m_) {
      s->comm_wait(kXHaloU);
      s->comm_wait(kXSums);
      if (fused)
        launch_cgu_tile(0, s->gu_, strided3(u_p(s), s->ucs_), strided3(u_w(s), s->ucs_), strided3(u_p(s), s->ucs_),
                        strided3(u_w(s), s->ucs_), strided3(s->su_alt_, s->ucs_), strided3(s->cgs_ra_, s->ucs_), Z, s->dscal_, 0, 0,
                        nglobal, small, s->partial_, s->dscal_ + s0, s->stream_);
      else
        launch_cgu_dot(0, s->gu_, strided3(u_p(s), s->ucs_), strided3(u_w(s), s->ucs_), Z, s->partial_, s->dscal_ + s0, s->stream_);
      s->slab_reduce(s0, 1, false);   // p : (p - w)
    }
    for (Solver* s : m_) {
      s->comm_wait(kXSums);
      // eps += alpha p ; r -= alpha (p - w),  alpha = gamma / (p:(p - w) / N + tiny); spare planes included
      if (fused) {
        launch_cgu_tile(1, s->gu_, strided3(u_e(s), s->ucs_), strided3(u_r(s), s->ucs_), strided3(u_p(s), s->ucs_),
                        strided3(u_w(s), s->ucs_), strided3(s->su_alt_, s->ucs_), strided3(s->cgs_ra_, s->ucs_), E, s->dscal_,
                        blk[cur] + 6, s0, nglobal, small, s->partial_, s->dscal_ + blk[nxt], s->stream_);
        launch_cgu_axpy_oop(0, strided3(u_e(s), s->ucs_), strided3(u_p(s), s->ucs_), strided3(u_r(s), s->ucs_),
                            strided3(u_w(s), s->ucs_), strided3(s->su_alt_, s->ucs_), strided3(s->cgs_ra_, s->ucs_), s->dscal_,
                            blk[cur] + 6, s0, nglobal, small, s->g_.n, s->ucs_ - s->g_.n, s->stream_);
        std::swap(s->su_[s->su_cur_], s->su_alt_);
        std::swap(s->cgs_r_, s->cgs_ra_);
      } else {
        launch_cgu_axpy(0, s->gu_, strided3(u_e(s), s->ucs_), strided3(u_p(s), s->ucs_), strided3(u_r(s), s->ucs_),
                        strided3(u_w(s), s->ucs_), s->dscal_, blk[cur] + 6, s0, nglobal, small, s->stream_, s->ucs_);
        launch_cgu_dot(1, s->gu_, strided3(u_e(s), s->ucs_), strided3(u_r(s), s->ucs_), E, s->partial_, s->dscal_ + blk[nxt], s->stream_);
      }
      s->slab_reduce(blk[nxt], 7, false);   // norms of eps ; r : r
    }
    fetch7(blk[nxt]);
    if (!voting && iter < a.opt_.maxiter) {   // the next direction and operator application, enqueued behind the copies
      if (fused_dir) {
        apply_dir(cur, nxt);
      } else {
        direction_update(cur, nxt);
        apply(true, Z.v);
      }
      applied = true;
    }
    wait_norms();
    for (Solver* s : m_) {   // state for accessors called from the callback / bc_error: eps = E + grad_s u_e
      s->su_valid_ = true;
      s->eps_stale_ = true;
      for (int c = 0; c < 6; ++c) s->E_cur_[c] = E.v[c];
    }
    double m[6], s9 = 0.0;
    for (int c = 0; c < 6; ++c) {
      const double ss = a.hscal_[kSlotCg + c];
      for (Solver* s : m_) s->sumsq_[c] = ss;
      m[c] = std::sqrt(ss / nglobal);
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
      gamma_cur = a.hscal_[kSlotCg + 6] / nglobal + small;
    }
    if (std::isnan(rel_err) || stop_requested()) {   // _converged  F:21177-21244, decisions on reduced values only
      failed = true;
      break;
    }
    for (Solver* s : m_) s->residuals_.push_back(rel_err);
    bool stop = false, cancelled = false;
    for (Solver* s : m_) {
      if (s->cb_ && s->cb_(s->cb_user_)) stop = true;
      if (s->cancel_) cancelled = true;
    }
    if (voting) {
      double v[2] = {stop ? 1.0 : 0.0, cancelled ? 1.0 : 0.0};
      vote(v);
      stop = v[0] != 0.0;
      cancelled = v[1] != 0.0;
    }
    if (a.nranks_ > 1 && !voting) cancelled = false;
    if (stop) break;
    if (cancelled) {
      failed = true;
      break;
    }
    if (iter >= a.opt_.maxiter) break;
    if (rel_err <= a.opt_.tol || abs_err <= a.opt_.abs_tol) {
      if (bc_error(E0, S0) <= a.opt_.bc_tol) break;
    }
    iter++;
    if (!applied) {
      if (fused_dir) dir_cur = cur, dir_nxt = nxt;   // formed inside the next operator application
      else direction_update(cur, nxt);
    }
  }
  for (Solver* s : m_) {
    s->in_run_ = false;
    s->iterations_ = iter;
    s->su_valid_ = true;
    s->eps_stale_ = true;
    for (int c = 0; c < 6; ++c) s->E_cur_[c] = E.v[c];
    s->slab_materialise_eps();
  }
  synchronize();
  const double dt = now_seconds() - t_start;
  for (Solver* s : m_) s->solve_time_ += dt;
  return failed;
}

// The same CG for the scalar modes, carried in potential space like Solver::run_cg_scalar (g = E + grad T_e; r, p, w = grad T_r,
// T_p, T_w; runCG sends every non-hyperelastic mode to runCGElasticity F:22056-22066, whose inner product is the plain sum for
// 3 components F:20961-20980).  T_e = component 0 of su_[cur], T_w = component 0 of su_[cur ^ 1] (the chain's output), T_r, T_p =
// scg_.  The point-wise updates run over the spare planes too, so only T_w needs its halo exchange; alpha and beta on the host.
bool SlabGroup::run_cg_scalar(const double* E0, double prev0) {
  Solver& a = *m_[0];
  const double t_start = now_seconds();
  const bool residual_est = a.opt_.error_estimator == 1;
  const double small = std::numeric_limits<double>::min();
  const double nglobal = (double)a.nglobal_;
  if (a.opt_.update_ref) {
    calc_ref_material();
    prepare();
  }
  const bool voting = agree_on_voting();
  Vec6 E, Z;
  for (int i = 0; i < 6; ++i) E.v[i] = i < 3 ? E0[i] : 0.0, Z.v[i] = 0.0;
  auto T_e = [](Solver* s) { return s->su_[s->su_cur_]; };
  auto T_w = [](Solver* s) { return s->su_[s->su_cur_ ^ 1]; };
  auto T_r = [](Solver* s) { return s->scg_; };
  auto T_p = [](Solver* s) { return s->scg_ + s->ucs_; };
  auto apply = [&](bool from_p, const double* Eadd) {   // T_w = operator(T_in) with prescribed mean gradient Eadd
    for (Solver* s : m_) s->slab_front_fast(Eadd, false, from_p ? T_p(s) : T_e(s), false);
    pass_fast_chain();
  };
  auto dot = [&](int mode, bool first_is_p, bool second_is_w, int slot, int n) {
    for (Solver* s : m_) {
      s->comm_wait(kXHaloU);
      s->comm_wait(kXSums);
      launch_sc_cg_dot(mode, s->gu_, first_is_p ? T_p(s) : T_e(s), second_is_w ? T_w(s) : T_r(s), mode == 1 ? E : Z, s->partial_,
                       s->dscal_ + slot, s->stream_);
    }
    reduce_and_fetch(slot, n, false);
  };
  for (Solver* s : m_) {
    s->slab_cg_alloc();
    s->comm_wait(kXHaloU);
    FG_HIP_CHECK(hipMemsetAsync(T_e(s), 0, 3 * (size_t)s->ucs_ * sizeof(double), s->stream_));   // g_0 = E
    s->su_valid_ = true;
    s->eps_stale_ = true;
    s->in_run_ = true;
    for (int i = 0; i < 6; ++i) s->E_cur_[i] = E.v[i];
  }
  apply(false, E.v);
  for (Solver* s : m_) {
    s->comm_wait(kXHaloU);
    const size_t f1 = (size_t)s->ucs_ * sizeof(double);
    FG_HIP_CHECK(hipMemcpyAsync(T_r(s), T_w(s), f1, hipMemcpyDeviceToDevice, s->stream_));
    FG_HIP_CHECK(hipMemcpyAsync(T_p(s), T_w(s), f1, hipMemcpyDeviceToDevice, s->stream_));   // p = r
  }
  // Fused form (option cg_fused; see Solver::run_cg_scalar and SlabGroup::run_cg): two tiled sweeps for the vector work, the
  // direction update inside the operator's sweep, the CG scalars on the device (all-reduced there), the next operator
  // application enqueued before the hosts wait for the sums.  Alternates: T_e in component 1 of its buffer (copied back to
  // component 0 at the end), T_r / T_p in components 2, 3 of scg_.  Not with convergence callbacks (accessors read component 0).
  bool fused = a.opt_.cg_fused != 0 && !voting;
  for (Solver* s : m_) fused = fused && sc_sweep_tiled(s->gu_);
  if (a.nranks_ > 1) {   // the two forms issue different reductions: every rank takes the form all of them can take
    double v[2] = {fused ? 0.0 : 1.0, 0.0};
    vote(v);
    fused = v[0] == 0.0;
  }
  if (fused) {
    const int blk[2] = {kSlotCg, kSlotCg + 8}, s0 = kSlotCg + 16;
    auto fetch7 = [&](int slot) {
      for (Solver* s : m_) {
        FG_HIP_CHECK(hipMemcpyAsync(s->hscal_ + kSlotCg, s->dscal_ + slot, 7 * sizeof(double), hipMemcpyDeviceToHost, s->comm_stream_));
        FG_HIP_CHECK(hipMemcpyAsync(s->hscal_ + kSlotFlag, s->dscal_ + kSlotFlag, 2 * sizeof(double), hipMemcpyDeviceToHost,
                                    s->comm_stream_));
        FG_HIP_CHECK(hipMemcpyAsync(s->herr_, s->derr_, sizeof(int), hipMemcpyDeviceToHost, s->comm_stream_));
        FG_HIP_CHECK(hipEventRecord(s->ev_norm_, s->comm_stream_));
      }
    };
    for (Solver* s : m_) {
      s->cgs_r_ = s->scg_;
      s->cgs_p_ = s->scg_ + s->ucs_;
      s->cgs_ra_ = s->scg_ + 2 * s->ucs_;
      s->cgs_pa_ = s->scg_ + 3 * s->ucs_;
    }
    std::vector<double*> e_cur(m_.size()), e_alt(m_.size());
    for (size_t i = 0; i < m_.size(); ++i) e_cur[i] = T_e(m_[i]), e_alt[i] = T_e(m_[i]) + m_[i]->ucs_;
    auto apply_dir = [&](int cur, int nxt) {
      for (Solver* s : m_) {
        s->comm_wait(kXSums);
        s->slab_front_fast_sc_cg(blk[nxt] + 6, blk[cur] + 6, nglobal, small);
      }
      pass_fast_chain();
    };
    for (size_t i = 0; i < m_.size(); ++i) {
      Solver* s = m_[i];
      s->comm_wait(kXHaloU);
      s->comm_wait(kXSums);
      launch_sc_cg_dot(1, s->gu_, e_cur[i], s->cgs_r_, E, s->partial_, s->dscal_ + blk[0], s->stream_);
      s->slab_reduce(blk[0], 7, false);   // gamma_0 = r.r / N + tiny
    }
    fetch7(blk[0]);
    wait_norms();
    double gamma_cur = a.hscal_[kSlotCg + 6] / nglobal + small;
    const double gamma_0 = gamma_cur;
    double prev = prev0;
    long iter = 0;
    bool failed = false, applied = false;
    for (;;) {
      const int cur = (int)(iter & 1), nxt = cur ^ 1;
      if (!applied) {   // the first iteration: p = r
        for (Solver* s : m_) s->slab_front_fast(Z.v, false, s->cgs_p_, false);
        pass_fast_chain();
      }
      applied = false;
      for (size_t i = 0; i < m_.size(); ++i) {
        Solver* s = m_[i];
        s->comm_wait(kXHaloU);
        s->comm_wait(kXSums);
        launch_sc_cgu_tile(0, s->gu_, s->cgs_p_, T_w(s), s->cgs_p_, T_w(s), e_alt[i], s->cgs_ra_, Z, s->dscal_, 0, 0, nglobal, small,
                           s->partial_, s->dscal_ + s0, s->stream_);
        s->slab_reduce(s0, 1, false);   // p . (p - w)
      }
      for (size_t i = 0; i < m_.size(); ++i) {
        Solver* s = m_[i];
        s->comm_wait(kXSums);
        launch_sc_cgu_tile(1, s->gu_, e_cur[i], s->cgs_r_, s->cgs_p_, T_w(s), e_alt[i], s->cgs_ra_, E, s->dscal_, blk[cur] + 6, s0, nglobal,
                           small, s->partial_, s->dscal_ + blk[nxt], s->stream_);
        launch_sc_cg_axpy_oop(0, e_cur[i], s->cgs_p_, s->cgs_r_, T_w(s), e_alt[i], s->cgs_ra_, s->dscal_, blk[cur] + 6, s0, nglobal,
                              small, s->g_.n, s->ucs_ - s->g_.n, s->stream_);
        std::swap(e_cur[i], e_alt[i]);
        std::swap(s->cgs_r_, s->cgs_ra_);
        s->slab_reduce(blk[nxt], 7, false);   // norms of g ; r . r
      }
      fetch7(blk[nxt]);
      if (iter < a.opt_.maxiter) {   // the next direction and operator application, enqueued behind the copies
        apply_dir(cur, nxt);
        applied = true;
      }
      wait_norms();
      double s3 = 0.0;
      for (int c = 0; c < 6; ++c) {
        const double ss = c < 3 ? a.hscal_[kSlotCg + c] : 0.0;
        for (Solver* s : m_) s->sumsq_[c] = ss;
        s3 += ss / nglobal;
      }
      const double curn = std::sqrt(s3);
      double abs_err = std::fabs(prev - curn);
      double rel_err = abs_err / (small + curn);
      prev = curn;
      if (residual_est) {   // update_cg(gamma, gamma0)  F:14397-14401 with the gamma this iteration started from
        abs_err = std::sqrt(gamma_cur);
        rel_err = std::sqrt(gamma_cur / gamma_0);
      }
      gamma_cur = a.hscal_[kSlotCg + 6] / nglobal + small;
      if (std::isnan(rel_err) || stop_requested()) {
        failed = true;
        break;
      }
      for (Solver* s : m_) s->residuals_.push_back(rel_err);
      if (iter >= a.opt_.maxiter) break;
      if (rel_err <= a.opt_.tol || abs_err <= a.opt_.abs_tol) {
        // bc_error reads the gradient of the current iterate: component 0 of the state buffer must be it
        for (size_t i = 0; i < m_.size(); ++i) {
          Solver* s = m_[i];
          if (e_cur[i] != T_e(s)) {
            FG_HIP_CHECK(hipMemcpyAsync(T_e(s), e_cur[i], (size_t)s->ucs_ * sizeof(double), hipMemcpyDeviceToDevice, s->stream_));
            std::swap(e_cur[i], e_alt[i]);
          }
          s->su_valid_ = true;
          s->eps_stale_ = true;
          for (int c = 0; c < 6; ++c) s->E_cur_[c] = E.v[c];
        }
        double S0[6] = {0, 0, 0, 0, 0, 0};
        if (bc_error(E.v, S0) <= a.opt_.bc_tol) break;
      }
      iter++;
    }
    for (size_t i = 0; i < m_.size(); ++i) {
      Solver* s = m_[i];
      if (e_cur[i] != T_e(s))
        FG_HIP_CHECK(hipMemcpyAsync(T_e(s), e_cur[i], (size_t)s->ucs_ * sizeof(double), hipMemcpyDeviceToDevice, s->stream_));
      s->in_run_ = false;
      s->iterations_ = iter;
      s->su_valid_ = true;
      s->eps_stale_ = true;
      for (int c = 0; c < 6; ++c) s->E_cur_[c] = E.v[c];
      s->slab_materialise_eps();
    }
    synchronize();
    const double dt = now_seconds() - t_start;
    for (Solver* s : m_) s->solve_time_ += dt;
    return failed;
  }
  dot(1, false, false, kSlotCg, 7);
  double gamma = a.hscal_[kSlotCg + 6] / nglobal + small;
  const double gamma_0 = gamma;
  double prev = prev0;
  long iter = 0;
  bool failed = false;
  for (;;) {
    apply(true, Z.v);                       // T_w = operator(T_p)
    dot(0, true, true, kSlotCg + 8, 1);     // p : (p - w)
    const double alpha = gamma / (a.hscal_[kSlotCg + 8] / nglobal + small);
    for (Solver* s : m_)   // g += alpha p ; r -= alpha (p - w), spare planes included
      launch_sc_cg_axpy(0, s->gu_, T_e(s), T_p(s), T_r(s), T_w(s), alpha, s->stream_, s->ucs_);
    dot(1, false, false, kSlotCg, 7);       // norms of g ; r : r
    const double rr = a.hscal_[kSlotCg + 6];
    for (Solver* s : m_) {
      s->su_valid_ = true;
      s->eps_stale_ = true;
      for (int c = 0; c < 6; ++c) s->E_cur_[c] = E.v[c];
    }
    double s3 = 0.0;
    for (int c = 0; c < 6; ++c) {
      const double ss = c < 3 ? a.hscal_[kSlotCg + c] : 0.0;
      for (Solver* s : m_) s->sumsq_[c] = ss;
      s3 += ss / nglobal;
    }
    const double cur = std::sqrt(s3);
    double abs_err = std::fabs(prev - cur);
    double rel_err = abs_err / (small + cur);
    prev = cur;
    if (residual_est) {   // update_cg(gamma, gamma0)  F:14397-14401
      abs_err = std::sqrt(gamma);
      rel_err = std::sqrt(gamma / gamma_0);
    }
    if (std::isnan(rel_err) || stop_requested()) {
      failed = true;
      break;
    }
    for (Solver* s : m_) s->residuals_.push_back(rel_err);
    bool stop = false, cancelled = false;
    for (Solver* s : m_) {
      if (s->cb_ && s->cb_(s->cb_user_)) stop = true;
      if (s->cancel_) cancelled = true;
    }
    if (voting) {
      double v[2] = {stop ? 1.0 : 0.0, cancelled ? 1.0 : 0.0};
      vote(v);
      stop = v[0] != 0.0;
      cancelled = v[1] != 0.0;
    }
    if (a.nranks_ > 1 && !voting) cancelled = false;
    if (stop) break;
    if (cancelled) {
      failed = true;
      break;
    }
    if (iter >= a.opt_.maxiter) break;
    if (rel_err <= a.opt_.tol || abs_err <= a.opt_.abs_tol) {
      double S0[6] = {0, 0, 0, 0, 0, 0};
      if (bc_error(E.v, S0) <= a.opt_.bc_tol) break;
    }
    iter++;
    const double delta = rr / nglobal + small;
    const double beta = delta / gamma;
    gamma = delta;
    for (Solver* s : m_) launch_sc_cg_axpy(1, s->gu_, T_e(s), T_p(s), T_r(s), T_w(s), beta, s->stream_, s->ucs_);   // p = r + beta p
  }
  for (Solver* s : m_) {
    s->in_run_ = false;
    s->iterations_ = iter;
    s->su_valid_ = true;
    s->eps_stale_ = true;
    for (int c = 0; c < 6; ++c) s->E_cur_[c] = E.v[c];
    s->slab_materialise_eps();
  }
  synchronize();
  const double dt = now_seconds() - t_start;
  for (Solver* s : m_) s->solve_time_ += dt;
  return failed;
}

// runCGElasticity  F:23153-23247 on the slabs in strain space: eps (in eps_), r, p, w as 6-component fields of the slab, the
// Krylov operator eps -> -Gamma0 (C - C0) eps (krylovOperator F:20583-20587: one basicScheme pass with E = 0, mixed-BC
// projector included) as one pass of the strain-state pipeline on a work field, the inner products all-reduced, alpha and
// beta on the host -- the collective counterpart of Solver::run_cg.  Any configuration the slab driver runs.
bool SlabGroup::run_cg_strain(const double* E0, const double* S0, double prev0) {
  Solver& a = *m_[0];
  const double t_start = now_seconds();
  const bool residual_est = a.opt_.error_estimator == 1;
  const double small = std::numeric_limits<double>::min();
  const double nglobal = (double)a.nglobal_;
  for (Solver* s : m_) {
    FG_HIP_CHECK(hipSetDevice(s->device_));
    const size_t f6 = 6 * (size_t)s->g_.n * sizeof(double);
    for (double** b : {&s->cg_r_, &s->cg_p_, &s->cg_w_})
      if (!*b) FG_HIP_CHECK(hipMalloc(b, f6));
    s->in_run_ = true;
  }
  if (a.opt_.update_ref) calc_ref_material();
  const bool voting = agree_on_voting();
  const bool mixed_bc = !(frobenius(a.BC_MQ_) < kEps);
  Vec6 E, Z;
  {
    double t1[6], t2[6], t3[6];   // calcBCMean  F:20242-20245
    voigt_mv(a.BC_QC0_, E0, t1);
    for (int i = 0; i < 6; ++i) t2[i] = S0[i] - t1[i];
    voigt_mv(a.BC_M_, t2, t3);
    for (int i = 0; i < 6; ++i) E.v[i] = E0[i] + a.opt_.bc_relax * t3[i], Z.v[i] = 0.0;
  }
  // dst = operator(src): the pass works in place on eps_, so the field is copied into dst and dst takes eps_'s place
  auto apply = [&](double* Solver::*src, double* Solver::*dst) {
    for (Solver* s : m_) {
      s->comm_wait(kXHaloU);
      FG_HIP_CHECK(hipMemcpyAsync(s->*dst, s->*src, 6 * (size_t)s->g_.n * sizeof(double), hipMemcpyDeviceToDevice, s->stream_));
      std::swap(s->eps_, s->*dst);
      s->su_valid_ = false;
      s->eps_stale_ = false;
    }
    try {
      pass_exact(Z.v, mixed_bc);
    } catch (...) {
      for (Solver* s : m_) std::swap(s->eps_, s->*dst);   // eps_ is the iterate again before the error leaves
      throw;
    }
    for (Solver* s : m_) {
      std::swap(s->eps_, s->*dst);
      s->su_valid_ = false;   // the displacement the pass left belongs to the work field, not to eps_
      s->eps_stale_ = false;
    }
  };
  auto dot = [&](int mode, double* Solver::*x, double* Solver::*y, double* Solver::*z, double alpha, int slot, int n) {
    for (Solver* s : m_)
      launch_cg(mode, s->g_, s->ptrs6(s->*x), s->ptrs6(s->*y), s->ptrs6(s->*z), E, alpha, s->partial_, s->dscal_ + slot, s->stream_);
    if (n > 0) reduce_and_fetch(slot, n, false);
  };
  for (Solver* s : m_) {
    s->comm_wait(kXHaloU);
    launch_set_const6(s->g_, s->ptrs6(s->eps_), E, s->stream_);   // eps_0 = E
    s->su_valid_ = false;
    s->eps_stale_ = false;
    for (int i = 0; i < 6; ++i) s->E_cur_[i] = E.v[i];
  }
  apply(&Solver::eps_, &Solver::cg_r_);                                          // r = -Gamma0 (C - C0) eps
  dot(0, &Solver::cg_r_, &Solver::eps_, &Solver::eps_, 0.0, kSlotMean, 1);       // r += E - eps ; r:r
  double gamma = a.hscal_[kSlotMean] / nglobal + small;
  const double gamma_0 = gamma;
  for (Solver* s : m_)
    FG_HIP_CHECK(hipMemcpyAsync(s->cg_p_, s->cg_r_, 6 * (size_t)s->g_.n * sizeof(double), hipMemcpyDeviceToDevice, s->stream_));
  double prev = prev0;
  long iter = 0;
  bool failed = false;
  for (;;) {
    apply(&Solver::cg_p_, &Solver::cg_w_);                                       // w = -Gamma0 (C - C0) p
    dot(1, &Solver::cg_p_, &Solver::cg_w_, &Solver::cg_w_, 0.0, kSlotMean, 1);   // p:(p - w)
    const double alpha = gamma / (a.hscal_[kSlotMean] / nglobal + small);
    dot(2, &Solver::eps_, &Solver::cg_p_, &Solver::cg_p_, alpha, kSlotSumSq, 6);  // eps += alpha p ; norms
    double m[6], s9 = 0.0;
    for (int c = 0; c < 6; ++c) {
      const double ss = a.hscal_[kSlotSumSq + c];
      for (Solver* s : m_) s->sumsq_[c] = ss;
      m[c] = std::sqrt(ss / nglobal);
    }
    for (int c = 0; c < 6; ++c) s9 += m[c] * m[c];
    for (int c = 3; c < 6; ++c) s9 += m[c] * m[c];
    const double cur = std::sqrt(s9);
    double abs_err = std::fabs(prev - cur);
    double rel_err = abs_err / (small + cur);
    prev = cur;
    if (residual_est) {   // update_cg(gamma, gamma0)  F:14397-14401
      abs_err = std::sqrt(gamma);
      rel_err = std::sqrt(gamma / gamma_0);
    }
    if (a.opt_.error_estimator >= 2) estimator_update(&abs_err, &rel_err);   // update_cg -> update  F:14465, F:14584
    if (std::isnan(rel_err) || stop_requested()) {
      failed = true;
      break;
    }
    for (Solver* s : m_) s->residuals_.push_back(rel_err);
    bool stop = false, cancelled = false;
    for (Solver* s : m_) {
      if (s->cb_ && s->cb_(s->cb_user_)) stop = true;
      if (s->cancel_) cancelled = true;
    }
    if (voting) {
      double v[2] = {stop ? 1.0 : 0.0, cancelled ? 1.0 : 0.0};
      vote(v);
      stop = v[0] != 0.0;
      cancelled = v[1] != 0.0;
    }
    if (a.nranks_ > 1 && !voting) cancelled = false;
    if (stop) break;
    if (cancelled) {
      failed = true;
      break;
    }
    if (iter >= a.opt_.maxiter) break;
    if (rel_err <= a.opt_.tol || abs_err <= a.opt_.abs_tol) {
      if (bc_error(E0, S0) <= a.opt_.bc_tol) break;
    }
    iter++;
    dot(3, &Solver::cg_r_, &Solver::cg_p_, &Solver::cg_w_, -alpha, kSlotMean, 1);   // r -= alpha (p - w) ; r:r
    const double delta = a.hscal_[kSlotMean] / nglobal + small;
    const double beta = delta / gamma;
    gamma = delta;
    dot(4, &Solver::cg_p_, &Solver::cg_r_, &Solver::cg_r_, beta, kSlotMean, 0);      // p = r + beta p
  }
  for (Solver* s : m_) {
    s->in_run_ = false;
    s->iterations_ = iter;
    s->su_valid_ = false;
    s->eps_stale_ = false;
  }
  synchronize();
  const double dt = now_seconds() - t_start;
  for (Solver* s : m_) s->solve_time_ += dt;
  return failed;
}

}  // namespace fg
