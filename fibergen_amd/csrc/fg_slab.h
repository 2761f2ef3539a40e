// Driver of the slab-decomposed Lippmann-Schwinger loop, below the C ABI.
//
// A SlabGroup holds the members this process drives: one (one process per GPU, RCCL or the callback transport) or all
// P of them (in-process group on one device).  Every collective operation is written as a sequence of steps executed
// "for step: for member", each step ending in at most one round of comm calls, so the same code serves both cases.
// The stop rule restates LSSolver::run / runBasic / _converged (F:21247-21398, F:21716-21805, F:21177-21244) exactly
// like Solver::run does for one GPU; the sums it needs are all-reduced on the device and reach every rank identically.
#pragma once

#include <vector>

#include "fg_solver.h"

namespace fg {

class SlabGroup {
 public:
  explicit SlabGroup(std::vector<Solver*> members, hipStream_t owned_stream = nullptr)
      : m_(std::move(members)), owned_stream_(owned_stream) {}
  ~SlabGroup() {
    if (owned_stream_) (void)hipStreamDestroy(owned_stream_);
  }
  SlabGroup(const SlabGroup&) = delete;
  SlabGroup& operator=(const SlabGroup&) = delete;
  const std::vector<Solver*>& members() const { return m_; }
  void set_members(std::vector<Solver*> members) { m_ = std::move(members); }
  void invalidate() { m_.clear(); }   // a member is being destroyed: the group must not be driven any more

  bool run(const double* E6, const double* S6);     // collective LSSolver::run (one load step); true = failed
  // runLoadsteppingSolver  F:21584-21685: steps first .. nparams-1 with params[i] * (E6, S6), each continuing from the one
  // before; basic scheme or CG (runCGElasticity F:23153-23247: displacement space where the fast path applies, strain
  // space otherwise)
  bool run_load_steps(const double* E6, const double* S6, const double* params, int nparams, int first, LoadstepCallback step_cb,
                      void* user);
  void iterate(const double* E6, int n);            // n passes without the stop rule (bench, profiling)
  void mean_stress(double* out6);
  double mean_energy();
  void estimator_begin(bool fresh);
  void estimator_update(double* abs_err, double* rel_err);
  void mean_strain(double* out6);
  double volume_fraction(int p);
  void calc_ref_material();
  void synchronize();

 private:
  void prepare();                                   // buffers, effective moduli + their halo planes
  bool fast_ok(bool allow_mixed_bc) const;
  void pass_fast(const double* E_cur, bool sum_tau, bool chain);       // steps 0..9 (chain: the speculative chain included)
  void pass_fast_chain();                                              // steps 1..9
  bool run_step(const double* E0, const double* S0, bool fresh);
  bool run_cg(const double* E0, const double* S0, bool fresh);
  bool run_cg_strain(const double* E0, const double* S0, double prev0);
  double current_norm9();
  void require_scalar_fast(bool allow_mixed_bc) const;
  bool run_cg_scalar(const double* E0, double prev0);
  bool agree_on_voting();                                              // does any rank carry a convergence callback?
  bool stop_requested() const;                                         // reduced flag word: some rank was cancelled
  void vote(double* v2);                                               // sums of two host values over the ranks
  void pass_exact(const double* E6, bool mixed_bc);                    // strain-state pipeline, adopts
  void wait_norms();
  void reduce_and_fetch(int slot, int n, bool min_op);                 // members' dscal_ slots -> reduced hscal_
  double bc_error(const double* E0, const double* S0);
  void check_members() const;
  std::vector<Solver*> m_;
  hipStream_t owned_stream_;   // the one stream of an in-process group (its members do not own it)
};

}  // namespace fg
