// extern "C" boundary of libfibergen_amd.so: converts exceptions to return codes.
#include <cstring>
#include <exception>
#include <string>

#include "../../include/fibergen_amd.h"
#include <chrono>
#include <memory>
#include <vector>

#include "fg_hip_util.h"
#include "fg_slab.h"
#include "fg_slab_plan.h"
#include "fg_fft_smooth_plans.h"
#include "fg_solver.h"

struct fg_solver {
  fg::Solver* impl;
  std::string error;
};

namespace {
thread_local std::string g_create_error;

template <class F>
int guarded(fg_solver* s, F&& f) {
  if (!s || !s->impl) return FG_ERROR;
  try {
    f(*s->impl);
    return FG_OK;
  } catch (const std::exception& e) {
    s->error = e.what();
  } catch (...) {
    s->error = "unknown error";
  }
  return FG_ERROR;
}
}  // namespace

// streaming copy / triad for fg_hbm_stream
__global__ void fg_stream_kernel(double2* a, const double2* b, const double2* c, long n2, int triad, double s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  double2 v = b[i];
  if (triad) {
    const double2 w = c[i];
    v.x += s * w.x;
    v.y += s * w.y;
  }
  a[i] = v;
}

extern "C" {

int fg_abi_version(void) { return FG_ABI_VERSION; }

const char* fg_last_error(const fg_solver* s) { return s ? s->error.c_str() : g_create_error.c_str(); }

fg_solver* fg_create(int nx, int ny, int nz, double dx, double dy, double dz, int device) {
  try {
    fg_solver* s = new fg_solver();
    try {
      s->impl = new fg::Solver(nx, ny, nz, dx, dy, dz, device);
    } catch (...) {
      delete s;
      throw;
    }
    return s;
  } catch (const std::exception& e) {
    g_create_error = e.what();
  } catch (...) {
    g_create_error = "unknown error";
  }
  return nullptr;
}

fg_solver* fg_create_slab(int nx, int ny, int nz, double dx, double dy, double dz, int device, int rank, int nranks) {
  try {
    fg_solver* s = new fg_solver();
    try {
      s->impl = new fg::Solver(nx, ny, nz, dx, dy, dz, device, rank, nranks, /*slab_layout=*/true);
    } catch (...) {
      delete s;
      throw;
    }
    return s;
  } catch (const std::exception& e) {
    g_create_error = e.what();
  } catch (...) {
    g_create_error = "unknown error";
  }
  return nullptr;
}

// ---- slab driver below the ABI ------------------------------------------------------------------------------------
int fg_comm_unique_id(char* id) {
  try {
    if (!id) throw std::runtime_error("NULL argument");
    fg::rccl_unique_id(id);
    return FG_OK;
  } catch (const std::exception& e) {
    g_create_error = e.what();
  }
  return FG_ERROR;
}

int fg_slab_connect_rccl(fg_solver* s, const char* id) {
  return guarded(s, [&](fg::Solver& v) {
    if (!id) throw std::runtime_error("NULL argument");
    if (!v.is_slab()) throw std::runtime_error("not a slab solver (create it with fg_create_slab)");
    if (v.has_group()) throw std::runtime_error("slab solver is already connected to a transport");
    auto comm = fg::make_rccl_comm(id, v.rank(), v.nranks(), v.device());
    v.connect(std::move(comm), std::make_shared<fg::SlabGroup>(std::vector<fg::Solver*>{&v}));
  });
}

int fg_slab_connect_callback(fg_solver* s, fg_exchange_fn exchange, fg_allreduce_fn allreduce, void* user) {
  return guarded(s, [&](fg::Solver& v) {
    auto comm = fg::make_callback_comm(v.rank(), v.nranks(), exchange, allreduce, user);
    v.connect(std::move(comm), std::make_shared<fg::SlabGroup>(std::vector<fg::Solver*>{&v}));
  });
}

const char* fg_slab_transport(const fg_solver* s) { return (s && s->impl) ? s->impl->transport() : ""; }

int fg_slab_group_create(int nx, int ny, int nz, double dx, double dy, double dz, int device, int nranks, fg_solver** out) {
  std::vector<fg_solver*> made;
  std::shared_ptr<fg::SlabGroup> group;   // outlives the members on the failure path: it owns their stream
  try {
    if (!out) throw std::runtime_error("NULL argument");
    if (nranks < 1 || nranks > 16) throw std::runtime_error("in-process slab group: 1..16 members");
    int ndev = 0;
    FG_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev < 1) throw std::runtime_error("no HIP device available: fibergen_amd needs an AMD GPU (gfx950)");
    if (device < 0 || device >= ndev) throw std::runtime_error("invalid device index");
    FG_HIP_CHECK(hipSetDevice(device));
    hipStream_t stream = nullptr;
    FG_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    group = std::make_shared<fg::SlabGroup>(std::vector<fg::Solver*>{}, stream);   // owns the stream
    auto hub = fg::make_local_hub(nranks);
    std::vector<fg::Solver*> members;
    for (int r = 0; r < nranks; ++r) {
      fg_solver* h = new fg_solver();
      made.push_back(h);
      h->impl = new fg::Solver(nx, ny, nz, dx, dy, dz, device, r, nranks, /*slab_layout=*/true, stream);
      members.push_back(h->impl);
    }
    group->set_members(members);
    for (int r = 0; r < nranks; ++r)
      members[r]->connect(nranks > 1 ? fg::make_local_comm(hub, r) : std::unique_ptr<fg::Comm>(), group);
    for (int r = 0; r < nranks; ++r) out[r] = made[r];
    return FG_OK;
  } catch (const std::exception& e) {
    g_create_error = e.what();
  } catch (...) {
    g_create_error = "unknown error";
  }
  for (fg_solver* h : made) {
    try {
      delete h->impl;
    } catch (...) {
    }
    delete h;
  }
  group.reset();   // destroys the shared stream, after the members that synchronise on it
  return FG_ERROR;
}

int fg_slab_plan(int nx, int ny, int nz, int nranks, int rank, int what, int comp, fg_plan_op* ops, int capacity,
                 fg_plan_op* self_copy) {
  if (nx < 1 || ny < 1 || nz < 1 || nranks < 1 || rank < 0 || rank >= nranks || nx % nranks || ny % nranks) return -1;
  if (what < FG_PLAN_A2A_FORWARD || what > FG_PLAN_HALO_TAU || comp < -1 || comp > 2) return -1;
  const fg::SlabPlan p = fg::slab_plan(fg::slab_dims(nx, ny, nz, nranks, rank), what, comp);
  if ((int)p.ops.size() > capacity && ops) return -1;
  if (ops)
    for (size_t i = 0; i < p.ops.size(); ++i) ops[i] = p.ops[i];
  if (self_copy) {
    self_copy[0] = p.self_src;
    self_copy[1] = p.self_dst;
  }
  return (int)p.ops.size();
}

void fg_destroy(fg_solver* s) {
  if (!s) return;
  try {
    delete s->impl;
  } catch (...) {
  }
  delete s;
}

int fg_set_num_phases(fg_solver* s, int n) {
  return guarded(s, [&](fg::Solver& v) { v.set_num_phases(n); });
}

int fg_set_phase(fg_solver* s, int p, double mu, double lambda, const double* phi) {
  return guarded(s, [&](fg::Solver& v) {
    v.set_phase_material(p, mu, lambda);
    if (phi) v.set_phase_field(p, phi);
  });
}

int fg_set_normals(fg_solver* s, const double* normals) {
  return guarded(s, [&](fg::Solver& v) {
    if (!normals) throw std::runtime_error("normals pointer is NULL");
    v.set_normals(normals);
  });
}

int fg_set_option_d(fg_solver* s, const char* key, double value) {
  return guarded(s, [&](fg::Solver& v) {
    const std::string k = key ? key : "";
    fg::SolverOptions& o = v.options();
    if (k == "tol") o.tol = value;
    else if (k == "abs_tol") o.abs_tol = value;
    else if (k == "bc_tol") o.bc_tol = value;
    else if (k == "ref_scale") o.ref_scale = value;
    else if (k == "bc_relax") o.bc_relax = value;
    else if (k == "mu_0") { o.mu_0 = value; v.reference_material_changed(); }
    else if (k == "lambda_0") { o.lambda_0 = value; v.reference_material_changed(); }
    else if (k == "eps_g") o.eps_g = value;
    else if (k == "eps_a") o.eps_a = value;
    else throw std::runtime_error("unknown option '" + k + "'");
  });
}

int fg_set_option_i(fg_solver* s, const char* key, long value) {
  return guarded(s, [&](fg::Solver& v) {
    const std::string k = key ? key : "";
    fg::SolverOptions& o = v.options();
    if (k == "maxiter") o.maxiter = value;
    else if (k == "mixing_rule") {
      if (value != FG_MIXING_VOIGT && value != FG_MIXING_LAMINATE) throw std::runtime_error("Unknown mixing rule");
      o.mixing = (int)value;
    } else if (k == "update_ref") o.update_ref = value != 0;
    else if (k == "mode") {
      if (value < 0 || value > 2) throw std::runtime_error("mode must be 0 (elasticity), 1 (heat / porous) or 2 (viscosity)");
      o.mode = (int)value;
      v.invalidate_moduli();   // the precomputed effective moduli depend on the mode's phase table
    }
    else if (k == "gamma_scheme") {
      if (value != 0 && value != 1) throw std::runtime_error("gamma_scheme must be 0 (staggered) or 1 (collocated)");
      o.gamma_scheme = (int)value;
    }
    else if (k == "u_tile") {
      o.u_tile = value != 0;
      v.invalidate_interface_lists();
    }
    else if (k == "fuse_x") o.fuse_x = value != 0;
    else if (k == "phi_sweep") o.phi_sweep = value != 0;
    else if (k == "laminate_overlap") o.laminate_overlap = value != 0;
    else if (k == "slab_loopback") o.slab_loopback = value != 0;
    else if (k == "slab_split") o.slab_split = value < 0 ? -1 : (value != 0);
    else if (k == "slab_interleave") o.slab_interleave = value < 0 ? -1 : (value != 0);
    else if (k == "cg_fused") o.cg_fused = value < 0 ? -1 : (value != 0);
    else if (k == "x_layout") o.x_layout = value < 0 ? -1 : (value != 0);
    else if (k == "plane_fft") o.plane_fft = value < 0 ? -1 : (value != 0);
    else if (k == "pair_chunk") o.pair_chunk = value < 0 ? 0 : (int)value;
    else if (k == "joint_x") o.joint_x = value != 0;
    else if (k == "tile_plans") {
      o.tile_plans = value != 0;
      fg::fft::smooth_plan_kernels(value != 0);
    }
    else if (k == "staged_copy") o.staged_copy = value < 0 ? -1 : (value != 0);
    else if (k == "stage_chunk_kb") {
      if (value < 1 || value > 16384) throw std::runtime_error("stage_chunk_kb must be 1 ... 16384");
      o.stage_chunk_kb = (int)value;
    }
    else if (k == "u_loop") o.u_loop = (int)value;
    else if (k == "method") {
      if (value != 0 && value != 1) throw std::runtime_error("Unknown solver method");
      o.method = (int)value;
    }
    else if (k == "fuse_stress_div") o.fuse_stress_div = value != 0;
    else if (k == "loadstep_extrapolation_order") {
      if (value < 0 || value > 7) throw std::runtime_error("loadstep_extrapolation_order must be 0..7");
      o.loadstep_extrapolation_order = (int)value;
    }
    else if (k == "error_estimator") {
      if (value < 0 || value > 4) throw std::runtime_error("Unknown error estimator (0 epsilon, 1 residual, 2 sigma, 3 energy, 4 none)");
      o.error_estimator = (int)value;
    }
    else throw std::runtime_error("unknown option '" + k + "'");
  });
}

int fg_set_bc_projector(fg_solver* s, const double* P36) {
  return guarded(s, [&](fg::Solver& v) {
    if (!P36) throw std::runtime_error("projector pointer is NULL");
    v.set_bc_projector(P36);
  });
}

int fg_set_convergence_callback(fg_solver* s, fg_callback cb, void* user) {
  return guarded(s, [&](fg::Solver& v) { v.set_callback(cb, user); });
}

int fg_cancel(fg_solver* s) {
  return guarded(s, [&](fg::Solver& v) { v.cancel(); });
}

int fg_run_load_case(fg_solver* s, const double* E6, const double* S6, int* failed) {
  return guarded(s, [&](fg::Solver& v) {
    if (!E6) throw std::runtime_error("strain pointer is NULL");
    const bool f = v.is_slab() ? v.slab_group().run(E6, S6) : v.run(E6, S6);   // slab solvers: collective call
    if (failed) *failed = f ? 1 : 0;
  });
}

int fg_run_load_steps(fg_solver* s, const double* E6, const double* S6, const double* params, int nparams, int first,
                      fg_loadstep_callback cb, void* user, int* failed) {
  return guarded(s, [&](fg::Solver& v) {
    if (!E6 || !params) throw std::runtime_error("NULL argument");
    const bool f = v.is_slab() ? v.slab_group().run_load_steps(E6, S6, params, nparams, first, cb, user)
                               : v.run_load_steps(E6, S6, params, nparams, first, cb, user);
    if (failed) *failed = f ? 1 : 0;
  });
}

int fg_iterate(fg_solver* s, const double* E6, int n) {
  return guarded(s, [&](fg::Solver& v) {
    if (!E6) throw std::runtime_error("strain pointer is NULL");
    if (v.is_slab()) v.slab_group().iterate(E6, n);
    else v.iterate(E6, n);
  });
}

int fg_time_iterations(fg_solver* s, const double* E6, int n, double* elapsed_ms) {
  return guarded(s, [&](fg::Solver& v) {
    if (!E6) throw std::runtime_error("strain pointer is NULL");
    if (v.is_slab()) {
      // exchanges run on a second stream: the timed region is bracketed on the host by full synchronisations
      fg::SlabGroup& grp = v.slab_group();
      grp.synchronize();
      const auto t0 = std::chrono::steady_clock::now();
      grp.iterate(E6, n);
      grp.synchronize();
      if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      return;
    }
    hipEvent_t a, b;
    FG_HIP_CHECK(hipEventCreate(&a));
    FG_HIP_CHECK(hipEventCreate(&b));
    FG_HIP_CHECK(hipEventRecord(a, v.stream()));
    v.iterate(E6, n);
    FG_HIP_CHECK(hipEventRecord(b, v.stream()));
    FG_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    FG_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (elapsed_ms) *elapsed_ms = ms;
  });
}

long fg_get_iterations(const fg_solver* s) { return (s && s->impl) ? s->impl->iterations() : 0; }

int fg_get_residuals(const fg_solver* s, double* out, int capacity) {
  if (!s || !s->impl) return 0;
  const std::vector<double>& r = s->impl->residuals();
  const int n = (int)r.size();
  if (out)
    for (int i = 0; i < n && i < capacity; ++i) out[i] = r[i];
  return n;
}

double fg_get_solve_time(const fg_solver* s) { return (s && s->impl) ? s->impl->solve_time() : 0.0; }

int fg_mean_stress(fg_solver* s, double* out6) {
  return guarded(s, [&](fg::Solver& v) {
    if (v.is_slab()) v.slab_group().mean_stress(out6);
    else v.mean_stress(out6);
  });
}
int fg_mean_strain(fg_solver* s, double* out6) {
  return guarded(s, [&](fg::Solver& v) {
    if (v.is_slab()) v.slab_group().mean_strain(out6);
    else v.mean_strain(out6);
  });
}
int fg_volume_fraction(fg_solver* s, int p, double* out) {
  return guarded(s, [&](fg::Solver& v) { *out = v.is_slab() ? v.slab_group().volume_fraction(p) : v.volume_fraction(p); });
}

int fg_calc_ref_material(fg_solver* s) {
  return guarded(s, [&](fg::Solver& v) {
    if (v.is_slab()) v.slab_group().calc_ref_material();
    else v.calc_ref_material();
  });
}
int fg_get_ref_material(const fg_solver* s, double* mu_0, double* lambda_0) {
  if (!s || !s->impl) return FG_ERROR;
  if (mu_0) *mu_0 = s->impl->mu_0();
  if (lambda_0) *lambda_0 = s->impl->lambda_0();
  return FG_OK;
}

int fg_field_components(const fg_solver* s, const char* name) {
  if (!s || !s->impl || !name) return 0;
  return s->impl->field_components(name);
}
int fg_get_field(fg_solver* s, const char* name, double* out) {
  return guarded(s, [&](fg::Solver& v) {
    if (!name || !out) throw std::runtime_error("NULL argument");
    v.get_field(name, out);
  });
}
int fg_set_field(fg_solver* s, const char* name, const double* in) {
  return guarded(s, [&](fg::Solver& v) {
    if (!name || !in) throw std::runtime_error("NULL argument");
    v.set_field(name, in);
  });
}

void* fg_device_pointer(fg_solver* s, const char* name, int comp) {
  if (!s || !s->impl || !name) return nullptr;
  return s->impl->device_component(name, comp);
}
void* fg_get_stream(fg_solver* s) { return (s && s->impl) ? (void*)s->impl->stream() : nullptr; }
int fg_synchronize(fg_solver* s) {
  return guarded(s, [&](fg::Solver& v) {
    if (v.is_slab()) v.slab_group().synchronize();
    else FG_HIP_CHECK(hipStreamSynchronize(v.stream()));
  });
}

int fg_run_stage(fg_solver* s, int stage, const double* E6) {
  return guarded(s, [&](fg::Solver& v) { v.run_stage(stage, E6); });
}
int fg_enable_stage_timing(fg_solver* s, int enable) {
  return guarded(s, [&](fg::Solver& v) { v.enable_stage_timing(enable != 0); });
}
int fg_get_stage_times(const fg_solver* s, double* ms, long* count) {
  if (!s || !s->impl) return FG_ERROR;
  const fg::StageTimes t = s->impl->stage_times();
  if (ms)
    for (int i = 0; i < fg::kNumTimedKernels; ++i) ms[i] = t.ms[i];
  if (count) *count = t.count;
  return FG_OK;
}

int fg_get_stage_timing_bias(const fg_solver* s, double* ms) {
  if (!s || !s->impl || !ms) return FG_ERROR;
  *ms = s->impl->event_bias_ms();
  return FG_OK;
}

long fg_get_counter(const fg_solver* s, const char* name) {
  if (!s || !s->impl || !name) return -1;
  return s->impl->counter(name);
}

int fg_get_comm_times(const fg_solver* s, double* ms) {
  if (!s || !s->impl || !ms) return FG_ERROR;
  s->impl->comm_times(ms);
  return FG_OK;
}

int fg_device_pci_bus_id(int device, char* out, int capacity) {
  try {
    if (!out || capacity < 16) throw std::runtime_error("fg_device_pci_bus_id: buffer of at least 16 bytes needed");
    FG_HIP_CHECK(hipDeviceGetPCIBusId(out, capacity, device));
    return FG_OK;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return FG_ERROR;
  }
}

int fg_hbm_stream(int device, int megabytes, int reps, double* copy_GBps, double* triad_GBps) {
  try {
    if (megabytes < 1 || reps < 1) throw std::runtime_error("fg_hbm_stream: megabytes and reps must be positive");
    FG_HIP_CHECK(hipSetDevice(device));
    const long n2 = (long)megabytes * 1024 * 1024 / 16;   // double2 elements per array
    double2 *a = nullptr, *b = nullptr, *c = nullptr;
    FG_HIP_CHECK(hipMalloc(&a, n2 * sizeof(double2)));
    FG_HIP_CHECK(hipMalloc(&b, n2 * sizeof(double2)));
    FG_HIP_CHECK(hipMalloc(&c, n2 * sizeof(double2)));
    FG_HIP_CHECK(hipMemset(b, 0, n2 * sizeof(double2)));
    FG_HIP_CHECK(hipMemset(c, 0, n2 * sizeof(double2)));
    hipEvent_t e0, e1;
    FG_HIP_CHECK(hipEventCreate(&e0));
    FG_HIP_CHECK(hipEventCreate(&e1));
    const unsigned nb = (unsigned)((n2 + 255) / 256);
    double best[2] = {0.0, 0.0};
    for (int mode = 0; mode < 2; ++mode)
      for (int r = 0; r < reps + 1; ++r) {   // first launch is a warm-up
        FG_HIP_CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fg_stream_kernel, dim3(nb), dim3(256), 0, 0, a, b, c, n2, mode, 0.5);
        FG_HIP_CHECK(hipEventRecord(e1, 0));
        FG_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        FG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double gbps = (mode ? 3.0 : 2.0) * (double)n2 * sizeof(double2) / (ms * 1e-3) / 1e9;
        if (r > 0 && gbps > best[mode]) best[mode] = gbps;
      }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    (void)hipFree(c);
    if (copy_GBps) *copy_GBps = best[0];
    if (triad_GBps) *triad_GBps = best[1];
    return FG_OK;
  } catch (const std::exception& e) {
    g_create_error = e.what();   // reported by fg_last_error(NULL)
    return FG_ERROR;
  }
}

}  // extern "C"
