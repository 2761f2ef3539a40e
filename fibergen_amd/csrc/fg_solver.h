// Single-GPU Lippmann-Schwinger solver (basic scheme, staggered grid, linear
// elasticity): the device-resident restatement of LSSolver<double,double,3>
// (F:14641-24740) for the path  run -> runLoadsteppingSolver -> runBasic ->
// basicScheme -> calcStress + GammaOperatorStaggered.
//
// All fields stay in HBM for the whole load case; per iteration only the six
// sums of squares of the error estimator cross back to the host.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "fg_comm.h"
#include "fg_fft.h"
#include "fg_hostmath.h"
#include "fg_kernels.h"
#include "fg_transfer.h"

namespace fg {

class SlabGroup;

// device scalar slots (dscal_ / hscal_)
namespace slots {
constexpr int kSlotSumSq = 0;    // 6
constexpr int kSlotMean = 6;     // 6
constexpr int kSlotMinMax = 12;  // 2
constexpr int kSlotMisc = 14;    // 2
constexpr int kSlotScratch = 16; // 6: sums nobody reads (strain materialisation)
constexpr int kSlotFlag = 22;    // 2: slab driver, summed over the ranks with every reduction: [0] ranks whose device error flag
                                 // is raised, [1] ranks that were asked to stop (fg_cancel)
constexpr int kSlotCg = 24;      // displacement CG: two blocks of 8 (norms of eps [6] + r:r, alternating per iteration), then p:(p-w) [8]
constexpr int kNumSlots = 48;
}  // namespace slots

typedef int (*ConvergenceCallback)(void* user);
typedef int (*LoadstepCallback)(void* user, int istep);

struct SolverOptions {
  // defaults of the LSSolver constructor  F:14800-14862
  double tol = 1e-4;
  double abs_tol = 2.220446049250313e-16;
  double bc_tol = 1e-3;
  long maxiter = 10000;
  double ref_scale = 1.0;
  double bc_relax = 1.0;
  int mixing = kMixVoigt;
  int update_ref = 1;           // update_ref != "never"
  double mu_0 = 0.0 / 1.0;      // set to NaN in the constructor unless <ref> given (F:15340)
  double lambda_0 = 0.0;
  double eps_g = 2.220446049250313e-16;          // laminate tolerances F:13110-13111
  double eps_a = 3.666852862501036e-11;          // eps^(2/3)
  int mode = 0;                 // 0 = elasticity, 1 = scalar (heat / porous: 3-component gradient, 1-component potential),
                                // 2 = viscosity (dual Stokes scheme: DeltaOperatorStaggered F:20422-20460, 6 components)
  int gamma_scheme = 0;         // 0 = staggered (GammaOperatorStaggered F:20288), 1 = collocated (GammaOperatorCollocated F:20302)
  int loadstep_extrapolation_order = 0;   // 0 = none, 1 = linear, ... (F:14696; method "polynomial" F:21468-21514)
  int slab_interleave = -1;     // -1: where available; 0: one message per peer and component
  int error_estimator = 0;      // 0 = epsilon (EpsilonErrorEstimator F:14591-14637), 1 = residual (ResidualErrorEstimator
                                // F:14382-14405: abs = sqrt(gamma), rel = sqrt(gamma / gamma_0); method cg only),
                                // 2 = sigma (F:14514-14587), 3 = energy (F:14410-14468), 4 = none (F:14370-14378)
  int method = 0;               // 0 = basic scheme (runBasic F:21716), 1 = conjugate gradients (runCGElasticity F:23153)
  int u_loop = 2;               // pure strain BC: displacement-based pass (0 off, 1 exact operation order: bit-identical
                                // to 0, 2 precomputed effective moduli + FMA, agrees with 1 to rounding; with laminate
                                // mixing 2 = the Voigt sweep + divergence of (tau_laminate - tau_voigt) at the interface)
  int fuse_stress_div = 1;      // Voigt mixing: polarisation + divergence in one sweep
  int fuse_x = 1;               // fuse x-FFT + Green operator + inverse x-FFT when the length allows
  int u_tile = 1;               // fast displacement sweep (u_loop = 2) as the LDS-tiled marching kernel where the grid allows
                                // (0 = the untiled sweep everywhere).  512^3: 2.8 -> 1.95 ms per sweep
  int phi_sweep = 1;            // two complementary phases: the tiled sweep reads phi_1 instead of the two moduli arrays
  int laminate_overlap = 1;     // displacement loop with laminate mixing: interface kernels on a second stream beside the sweep
  int slab_loopback = 0;        // test mode: a lone slab sends to itself through its transport (see Solver::slab_loopback)
  int slab_split = -1;          // slab driver: all-to-all per component, overlapping the next component's transforms (1), one
                                // exchange for the three components (0), or by slab size (-1)
  int plane_fft = -1;           // z and y transforms of a z-y plane in one kernel (small grids): -1 where available, 0 off, 1 on
  int x_layout = -1;            // x-contiguous intermediate layout [zc/8][y][x][8] between the y passes and the fused x pass
                                // (the spectrum goes through tau_, free in the displacement loop): 1 on, 0 off, -1 by size
  int cg_fused = -1;            // displacement-space CG with fused vector sweeps (run_cg_u): -1 where the tiled sweep fits, 0 off, 1 on
  int pair_chunk = 0;           // z and y passes of the transform chain in runs of this many x planes, r2c(c) -> y(c) and
                                // y^-1(c) -> c2r(c) (hand-over inside the Infinity Cache; measured slower, off by default)
  int staged_copy = -1;         // host <-> device field transfers through the pinned-buffer pipeline (fg_transfer.h): -1 for
                                // downloads of 8 MB and more, 1 always (both directions), 0 never (one strided copy)
  int joint_x = 1;              // tile kernels (lengths such as 100, 200, 300): the fused x pass of three components on ONE joint
                                // image (1) or on one image per component (0); identical butterflies, the default is the faster form
  int tile_plans = 1;           // tile kernels (decimal sizes): the kernels built for one plan each where the plan is in their tables
                                // (fg_fft_smooth_plans.h) (1) or the class kernels for every plan (0); process-wide switch, same butterflies
  int stage_chunk_kb = 16384;   // pipeline stage of the staged transfers (<= 16 MB; tests shrink it)
};

// The error estimators that re-measure a mean of the strain field every iteration (create_error_estimator F:14940-14972):
// SigmaErrorEstimator F:14514-14587 (created with _mode = 2: from its third update on, the mean of the distances to the
// last two mean stresses), EnergyErrorEstimator F:14410-14468, NoneErrorEstimator F:14370-14378 (always 1).  The solver
// (one GPU or the slab group) supplies the measurement; norm_2 runs over the 9 mirrored entries (fix_dim).
struct MeanEstimator {
  double m_prev[6] = {0, 0, 0, 0, 0, 0}, m_pp[6] = {0, 0, 0, 0, 0, 0}, w_prev = 0.0;
  long iter = 0;
  static double norm9_diff(const double* a, const double* b) {
    double s = 0.0;
    for (int c = 0; c < 6; ++c) s += (a[c] - b[c]) * (a[c] - b[c]) * (c >= 3 ? 2.0 : 1.0);
    return std::sqrt(s);
  }
  void start_sigma(const double* m) {
    for (int c = 0; c < 6; ++c) m_prev[c] = m_pp[c] = m[c];
    iter = 0;
  }
  void update_sigma(const double* m, double* abs_err, double* rel_err) {
    const double zero[6] = {0, 0, 0, 0, 0, 0};
    *abs_err = iter > 1 ? 0.5 * (norm9_diff(m_pp, m) + norm9_diff(m_prev, m)) : norm9_diff(m_prev, m);
    *rel_err = *abs_err / (std::numeric_limits<double>::min() + norm9_diff(m, zero));
    for (int c = 0; c < 6; ++c) m_pp[c] = m_prev[c], m_prev[c] = m[c];
    ++iter;
  }
  void start_energy(double w) { w_prev = w, iter = 0; }
  void update_energy(double w, double* abs_err, double* rel_err) {
    *abs_err = std::fabs(w_prev - w);
    *rel_err = *abs_err / (std::numeric_limits<double>::min() + std::fabs(w));
    w_prev = w;
    ++iter;
  }
};

enum Stage {
  kStageStress = 0,     // eps -> tau            calcStressDiff (mu_0, lambda_0)
  kStageDiv = 1,        // tau -> f
  kStageFftForward = 2, // f -> f_hat (scaled 1/N)
  kStageG0 = 3,         // f_hat -> u_hat (alpha = -1)
  kStageFftInverse = 4, // u_hat -> u
  kStageEps = 5,        // u -> eps (+E, +R), sums of squares
  kStageIteration = 6,  // all of the above = one basicScheme call
  kStageStressConst = 7 // eps -> tau = C0 : eps
};

// kernels of one pass, in launch order (HIP-event timing slots)
constexpr int kNumTimedKernels = 10;
//  0 stress  1 div  2 r2c_z  3 c2c_y_fwd  4 c2c_x_fwd  5 g0  6 c2c_x_inv  7 c2c_y_inv  8 c2r_z  9 eps_norm
struct StageTimes {
  double ms[kNumTimedKernels];
  long count;
};

class Solver {
 public:
  // (nx, ny, nz) is the GLOBAL grid; with nranks > 1 this object holds x-slab `rank` of it.
  // slab_layout: created through fg_create_slab / fg_slab_group_create (the loop then runs under the slab driver, also
  // for nranks = 1); shared_stream: all members of an in-process group enqueue on one stream (not owned).
  Solver(int nx, int ny, int nz, double dx, double dy, double dz, int device, int rank = 0, int nranks = 1,
         bool slab_layout = false, hipStream_t shared_stream = nullptr);
  ~Solver();
  Solver(const Solver&) = delete;
  Solver& operator=(const Solver&) = delete;

  const Grid& grid() const { return g_; }
  SolverOptions& options() { return opt_; }
  void invalidate_moduli() { mod_dirty_ = smod_dirty_ = complement_dirty_ = true; }
  void invalidate_interface_lists() { mixed_dirty_ = true; }   // which lists exist depends on u_tile
  void reference_material_changed() { recompute_bc(); }   // (mu_0, lambda_0) set from outside: M, MQ depend on C0
  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }

  void set_num_phases(int n);
  int num_phases() const { return pt_.n; }
  void set_phase_material(int p, double mu, double lambda);
  void set_phase_field(int p, const double* phi_host);  // [nx][ny][nz], copied
  void set_normals(const double* n_host);                // [3][nx][ny][nz]
  void set_bc_projector(const double* P36);              // row-major 6x6
  void set_callback(ConvergenceCallback cb, void* user) { cb_ = cb; cb_user_ = user; }
  void cancel() { cancel_ = true; }

  // LSSolver::run  F:21247-21398.  Returns false on success, true on error like the reference.
  bool run(const double* E6, const double* S6);
  // runLoadsteppingSolver  F:21584-21685: steps first .. nparams-1 with prescribed values params[i] * (E6, S6), each
  // continuing from the one before; step_cb after every step (non-zero = stop, reported as failure like the reference)
  bool run_load_steps(const double* E6, const double* S6, const double* params, int nparams, int first,
                      LoadstepCallback step_cb, void* user);
  // timed iterations without convergence logic (bench / profiling): n passes of basicScheme
  void iterate(const double* E6, int n);

  void mean_stress(double* out6);   // calcMeanStress  F:17793-17811
  double mean_energy();             // meanW  F:12239-12262 of the current strain field
  void mean_strain(double* out6);   // TensorField::average  F:10171
  double volume_fraction(int p);

  const std::vector<double>& residuals() const { return residuals_; }
  long iterations() const { return iterations_; }
  double solve_time() const { return solve_time_; }
  double mu_0() const { return opt_.mu_0; }
  double lambda_0() const { return opt_.lambda_0; }
  void calc_ref_material();         // calcRefMaterial  F:22283-22313

  // field access (host copies, z padding stripped / added)
  int field_components(const std::string& name) const;
  void get_field(const std::string& name, double* out_host);
  void set_field(const std::string& name, const double* in_host);
  // device pointer of a padded component (for zero-copy wrapping by the caller)
  double* device_component(const std::string& name, int c);

  int rank() const { return rank_; }
  int nranks() const { return nranks_; }
  int nx_global() const { return nxg_; }

  // slab driver below the ABI (fg_slab.hip): transport + the group of members this process drives
  bool is_slab() const { return slab_layout_; }
  void connect(std::unique_ptr<Comm> comm, std::shared_ptr<SlabGroup> group);
  bool has_group() const { return (bool)group_; }
  SlabGroup& slab_group();          // throws unless connected (a lone nranks = 1 slab connects to itself)
  const char* transport() const { return comm_ ? comm_->name() : (slab_layout_ && nranks_ == 1 ? "self" : ""); }

  // single stages on the solver's own buffers (parity tests, profiling)
  void run_stage(int stage, const double* E6);
  void enable_stage_timing(bool on);
  StageTimes stage_times() const { return times_; }
  // slab driver, while stage timing is on: milliseconds spent in the exchanges on the exchange stream (HIP events around
  // each one; the host waits for every exchange, so nothing overlaps while this is measured):
  // [0] all-to-all forward, [1] all-to-all backward, [2] halo planes, [3] all-reduces; same pass count as stage_times()
  void comm_times(double* ms4) const {
    for (int i = 0; i < 4; ++i) ms4[i] = comm_ms_[i];
  }
  void reset_stage_times();
  double event_bias_ms() const { return event_bias_ms_; }
  // fg_get_counter: "interface_voxels" / "affected_voxels" = lengths of the laminate correction's lists (0 before they are
  // built); -1 = unknown name
  long counter(const std::string& name) const;

 private:
  // one pass  dst = E - Gamma0 : (C - C0) : src  (defaults: the solver's strain field, in place)
  void basic_scheme(const double* E6, double* src = nullptr, double* dst = nullptr);
  void release();
  bool run_one_step(const double* E0, const double* S0);
  double current_norm9();
  // estimators 2 (sigma), 3 (energy), 4 (none): constructor on the field a step starts from, update after an iteration
  void estimator_begin(bool fresh);
  void estimator_update(double* abs_err, double* rel_err);
  MeanEstimator est_;
  bool run_cg(const double* E0, const double* S0, double prev0);
  bool run_cg_scalar(const double* E0, double prev0);  // heat / porous: CG in potential space
  bool run_cg_u(const double* E0, double prev0);      // the same CG carried in displacement space (Voigt, prescribed mean strains)
  bool u_loop_eligible(bool allow_mixed_bc = false) const;
  bool two_phase_complementary();       // phi_0 == 1 - phi_1 everywhere (checked once per geometry)
  FieldPtrs<2> effective_moduli();      // per-voxel sums of the phase moduli for the fast kernels (allocated on first use)
  void build_laminate_lists();          // interface / affected voxel lists of the laminate correction (once per geometry)
  void u_pass_front(const double* E6);  // u_k (fu_) -> sums of squares of eps_k, f_{k+1} (fu_alt_)
  void u_pass_back();                   // f_{k+1} -> u_{k+1}, buffers swapped
  // r2c, y, x + Green operator + x^-1, y^-1, c2r on 3 components
  // c12: optional {c10, c20} replacing the factors derived from (mu_0, lambda_0, alpha)
  // xscratch: three free components the spectrum may pass through in the x-contiguous layout (nullptr: in place, plain layout)
  bool plane_fft_on() const;
  void fft_g0_chain(double* buf, double alpha = -1.0, const double* c12 = nullptr, double* xscratch = nullptr);
  void ensure_eps();                    // materialise eps = E + sym grad u if the loop left it implicit
  void recompute_bc();
  double bc_error(const double* E_cur, const double* S_cur);
  StressParams stress_params(double mu_0, double lambda_0, double alpha) const;
  ScalarParams scalar_params(double mu_0, double alpha) const;
  FieldPtrs<kMaxPhases> phase_ptrs() const;
  PhaseTable phase_table() const;  // constants the kernels see (viscosity: Hooke constants equivalent to the scalar law)
  FieldPtrs<6> ptrs6(double* base) const;
  FieldPtrs<3> ptrs3(double* base) const;
  void check_device_error(const char* where);
  void fetch_norms_and_errors(const char* where);  // D2H of the sums of squares + error flag, waits for those copies only
  void launch_pending_back();
  void adopt_back();
  void upload_padded(double* dst, const double* src_unpadded, int ncomp = 1, long dstride = 0);
  void download_unpadded(const double* src, double* dst_unpadded, int ncomp = 1, long dstride = 0);
  bool staged_copy(size_t bytes, bool download) const;
  void upload_rows(const std::vector<RowBlock>& blocks, long len, long pitch);
  void download_rows(const std::vector<RowBlock>& blocks, long len, long pitch);
  int pair_chunk_planes(int ncomp) const;
  template <class A, class B>
  void run_pairs(int pc, A first, B second);


  void time_begin(int stage);
  void time_end(int stage);

  // ---- slab driver: per-member steps (fg_slab.hip); every comm call is the last dependent thing of its step
  friend class SlabGroup;
  void slab_alloc();
  bool slab_fast_ok(bool allow_mixed_bc) const;
  void slab_moduli_step();                               // effective moduli of the slab + exchange of their halo planes
  // su_[cur] (or u_src) -> norms of eps_k (all-reduced unless !reduce), f_{k+1} in fu_
  void slab_front_fast(const double* E6, bool sum_tau, const double* u_src = nullptr, bool reduce = true);
  void slab_front_laminate(bool sum_tau, bool reduce = true);
  void slab_cg_alloc();                                  // u_r, u_p of the displacement-space CG (with spare planes)
  bool slab_cg_alloc_fused();                            // + the alternate buffers of the fused sweeps; false: they do not fit
  // the tiled sweep on the new search direction u_p = u_r + beta u_p (formed on the fly, own planes -> cgs_pa_, spare planes
  // point-wise), Voigt mixing; then cgs_p_ / cgs_pa_ are swapped
  void slab_front_fast_cg(const double* E6, int i_num, int i_den, double nvox, double small);
  void slab_front_fast_sc_cg(int i_num, int i_den, double nvox, double small);   // the scalar modes' sibling (potential space)
  void slab_fetch_norms(int n);                          // D2H of the reduced sums (+ flag word), event for the host
  void comm_time_begin();
  void comm_time_end(int category);
  void slab_chain_step(int k);                           // k = 1..9: transform chain fu_ -> su_[next], see fg_slab.hip
  void slab_front_exact(bool sum_tau);                   // eps_ -> tau_, halo of tau (, sums of tau all-reduced)
  void slab_div_exact();                                 // tau_ + halo -> fu_
  void slab_back_exact(const double* E6, const double* R6);  // su_[next] + halo -> eps_, sums of squares all-reduced
  void slab_adopt(const double* E6, bool u_is_state);
  void slab_materialise_eps();                           // eps_ = E_cur + sym grad su_[cur]
  void slab_reset_state();
  void slab_reduce(int slot, int n, bool min_op);        // all-reduce of dscal_ slots on the comm stream
  void slab_exchange(int what, int comp, int done_slot);
  double* slab_buffer(int id);
  bool slab_loopback() const;
  bool slab_interleave() const;                          // batched exchanges: one message per peer (three components per block)
  bool slab_split() const;                               // all-to-all per component (overlap) or once for all three
  void comm_begin();
  void comm_end(int slot);
  void comm_wait(int slot);

  bool slab_layout_ = false;
  bool owns_stream_ = true;
  std::unique_ptr<Comm> comm_;
  std::shared_ptr<SlabGroup> group_;
  hipStream_t comm_stream_ = nullptr;   // exchanges (own stream with RCCL; the compute stream otherwise)
  bool owns_comm_stream_ = false;
  static constexpr int kCommSlots = 10; // 0-2 all-to-all forward, 3-5 backward, 6 halo of u, 7 halo of tau, 8 moduli, 9 sums
  hipEvent_t ev_c2x_ = nullptr;         // compute -> comm
  hipEvent_t ev_x_[kCommSlots] = {};    // comm -> compute
  bool x_pending_[kCommSlots] = {};
  hipEvent_t ev_norm_ = nullptr;        // the reduced sums have reached the host
  hipEvent_t ev_ct_[2] = {nullptr, nullptr};   // exchange timing (stage timing on)
  double comm_ms_[4] = {0, 0, 0, 0};
  double* su_[2] = {nullptr, nullptr};  // displacement of the slab, 3 components of ucs_ doubles (4 spare planes each)
  int su_cur_ = 0;
  bool su_valid_ = false;               // su_[su_cur_] (with valid halo planes) is the state: eps = E_cur_ + sym grad u
  double* scg_ = nullptr;               // displacement-space CG on slabs: u_r, u_p (3 components of ucs_ doubles each)
  // fused CG sweeps on slabs (out of place): the alternates of u_r, u_p (scg2_) and of u_e (su_alt_, swapped with su_[su_cur_]),
  // and which half of each pair is current
  double* scg2_ = nullptr;
  double* su_alt_ = nullptr;
  double *cgs_r_ = nullptr, *cgs_p_ = nullptr, *cgs_ra_ = nullptr, *cgs_pa_ = nullptr;
  double* smod_ = nullptr;              // effective moduli, 2 components of ucs_ doubles
  bool smod_dirty_ = true;
  bool slab_phi_ = false;               // smod_ component 0 holds phi_1 (two complementary phases), not the moduli
  long ucs_ = 0;
  Grid gu_;                             // g_ with the halo-plane mapping of the marching sweep
  std::unique_ptr<Fft3> fft_ys_;        // x pass + Green operator on the y-slab [nxg][nyl][nzc]

  Grid g_;
  SolverOptions opt_;
  PhaseTable pt_;
  int device_;
  hipStream_t stream_ = nullptr;
  int rank_ = 0, nranks_ = 1;
  int nxg_ = 0;        // global nx (g_.nx is the local slab thickness)
  int nyl_ = 0;        // ny / nranks: thickness of the y-slab after the transpose
  long nglobal_ = 0;   // global voxel count
  std::unique_ptr<Fft3> fft_;
  double* halo_[4] = {nullptr, nullptr, nullptr, nullptr};  // send_lo, send_hi, recv_lo, recv_hi (2 planes each)

  double* eps_ = nullptr;      // 6 padded components
  double* tau_ = nullptr;      // 6
  double* fu_ = nullptr;       // 3 (real f / u, complex f_hat / u_hat); after a pass it holds u
  double *cg_r_ = nullptr, *cg_p_ = nullptr, *cg_w_ = nullptr;  // CG residual, direction, operator image (6 each)
  double* fu_cg_ = nullptr;   // fused displacement-space CG: the alternate buffer of the iterate (3 components; swapped with fu_)
  unsigned* mixed_list_ = nullptr;  // element offsets of the interface voxels (laminate mixing, displacement loop)
  unsigned mixed_n_ = 0;
  bool mixed_dirty_ = true;
  unsigned* aff_list_ = nullptr;    // voxels whose divergence stencil touches an interface voxel
  int* aff_slots_ = nullptr;        // 8 per entry: interface-list index of self, x-1, x+1, y-1, y+1, z-1, z+1 (or -1)
  unsigned aff_n_ = 0;
  double* dtau_ = nullptr;          // [mixed_n_][6] tau_laminate - tau_voigt of the current pass
  // compact static copies for the interface solve (phase fractions [nph][n], normals [3][n]) and its strain scratch [6][n]
  double *lam_phic_ = nullptr, *lam_nrmc_ = nullptr, *lam_epsc_ = nullptr;
  double* mod_ = nullptr;      // 2: per-voxel effective moduli (sum phi 2 mu, sum phi lambda) of the fast sweep
  bool mod_dirty_ = true;
  bool complement_dirty_ = true, complementary_ = false;
  double* fu_alt_ = nullptr;   // 3: second f/u buffer of the displacement-based loop (swapped with fu_)
  double* phi_ = nullptr;      // nphase
  double* normals_ = nullptr;  // 3 (allocated on demand)
  double* partial_ = nullptr;  // reduction partials
  double* dscal_ = nullptr;    // device scalars
  double* hscal_ = nullptr;    // pinned host mirror
  int* derr_ = nullptr;        // device error flag
  int* herr_ = nullptr;
  unsigned* hseq_ = nullptr;   // pinned: sequence number the stop rule's publish kernel writes last
  unsigned publish_seq_ = 0;
  double* g0_kpm_[3] = {nullptr, nullptr, nullptr};
  cplx* g0_kp_[3] = {nullptr, nullptr, nullptr};
  double* xi_[3] = {nullptr, nullptr, nullptr};  // collocated scheme: signed frequency / cell size per axis

  hostmath::Mat6 BC_P_, BC_Q_, BC_M_, BC_MQ_, BC_QC0_;
  double F00_[6];
  ConvergenceCallback cb_ = nullptr;
  void* cb_user_ = nullptr;
  std::atomic<bool> cancel_{false};   // set by fg_cancel, possibly from another thread

  std::vector<double> residuals_;
  long iterations_ = 0;
  double solve_time_ = 0.0;
  double sumsq_[6];

  bool u_valid_ = false;    // fu_ holds the displacement belonging to the current strain state
  bool eps_stale_ = false;  // eps_ has not been written since fu_ changed
  bool in_run_ = false;
  bool fresh_step_ = true;   // the load step being run starts from the zeroed field (not from a previous step)
  bool cg_u_active_ = false;  // displacement-space CG is iterating: fu_ is the iterate u_e, fu_alt_ is free between steps
  double E_cur_[6] = {0, 0, 0, 0, 0, 0};   // prescribed strain the current (u, eps) state was built with
  double E_next_[6] = {0, 0, 0, 0, 0, 0};
  bool sc_tau_sums_ = false;   // the last scalar sweep left the sums of tau in kSlotMean (mixed BC, tiled sweep)
  bool timing_ = false;
  double event_bias_ms_ = 0.0;   // reading of an empty event pair, subtracted from every timed kernel
  StageTimes times_;
  hipEvent_t ev_[2] = {nullptr, nullptr};
  hipEvent_t ev_copy_ = nullptr;
  hipStream_t aux_stream_ = nullptr;    // second stream of the laminate correction (created on first use) + fork / join events
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  bool pending_back_ = false;  // run(): the sweep of this pass is enqueued, its FFT chain not yet
  bool back_ready_ = false;    // the FFT chain is enqueued (fu_alt_ will hold u_{k+1}) but not adopted
};

}  // namespace fg
