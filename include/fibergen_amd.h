/* fibergen_amd -- C ABI of the MI355X-native Lippmann-Schwinger solver.
 *
 * This is the drop-in boundary for fibergen's hot path.
 * The reference has no C ABI of its own: the path is reached through the
 * boost.python class FG (src/fibergen.cpp:27142-27187) and the C++ class
 * LSSolver<double,double,3> behind it.  Each entry point below names the
 * reference member it stands in for (F: = /root/reference/src/fibergen.cpp).
 * A reference-side binding would call these from FG<T,R,DIM>::run_actions
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - plain C: pointers and sizes only, no exceptions cross the boundary;
 *  - every function returning int returns FG_OK (0) or FG_ERROR (1); the
 *    message is available from fg_last_error();
 *  - host arrays are float64, C order [ncomp][nx][ny][nz] without z padding,
 *    component order 11,22,33,23,13,12 with plain tensor shear components;
 *  - all work is enqueued on the solver's own HIP stream; calls that return
 *    data synchronise that stream, nothing else.
 *  - not re-entrant per solver object (like the reference, F:380-406); distinct
 *    solver objects may be driven from distinct threads.
 */
#ifndef FIBERGEN_AMD_H
#define FIBERGEN_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define FG_OK 0
#define FG_ERROR 1
#define FG_ABI_VERSION 1

/* mixing_rule option values (F:15129, create_mixing_rule) */
#define FG_MIXING_VOIGT 0
#define FG_MIXING_LAMINATE 1

/* stages of one basic-scheme pass, for fg_run_stage (each is one reference routine) */
#define FG_STAGE_STRESS 0        /* calcStressDiff            F:18030-18033, F:18134-18184 : epsilon -> tau   */
#define FG_STAGE_DIV 1           /* divOperatorStaggered      F:18853-18908                : tau -> f         */
#define FG_STAGE_FFT_FORWARD 2   /* fftVector                 F:18481-18510                : f -> f_hat       */
#define FG_STAGE_G0 3            /* G0OperatorFourierStaggered F:19749-19755, F:19834-19927 : f_hat -> u_hat  */
#define FG_STAGE_FFT_INVERSE 4   /* fftInvVector              F:18513-18528                : u_hat -> u       */
#define FG_STAGE_EPS 5           /* epsOperatorStaggered      F:18614-18692 (+ component_norm F:10127) : u -> epsilon */
#define FG_STAGE_ITERATION 6     /* basicScheme               F:20558-20578                                    */
#define FG_STAGE_STRESS_CONST 7  /* calcStressConst           F:17973-18020                : epsilon -> tau   */

typedef struct fg_solver fg_solver;

/* convergence callback, consulted once per iteration before the tolerance test
 * (F:21215); return non-zero to stop.  Called on the caller's thread. */
typedef int (*fg_callback)(void* user);

int fg_abi_version(void);

/* Last error message of `s`, or (s == NULL) of the last failed fg_create on this thread. */
const char* fg_last_error(const fg_solver* s);

/* LSSolver::LSSolver(nx,ny,nz,dx,dy,dz)  F:14780-14892.  Allocates the strain field,
 * the polarisation/displacement work fields and the FFT plan on HIP device `device`.
 * Returns NULL on failure (no GPU, out of memory, bad sizes). */
fg_solver* fg_create(int nx, int ny, int nz, double dx, double dy, double dz, int device);
void fg_destroy(fg_solver* s);

/* Materials: the <materials> block of readSettings  F:15177-15299 after parameter conversion
 * (F:7333-7454).  Phase order = materials order.  phi may be NULL to keep the current field. */
int fg_set_num_phases(fg_solver* s, int nphases);
int fg_set_phase(fg_solver* s, int p, double mu, double lambda, const double* phi /* [nx][ny][nz] */);
/* interface normals for laminate mixing (LSSolver::_normals, F:14717) */
int fg_set_normals(fg_solver* s, const double* normals /* [3][nx][ny][nz] */);

/* Solver settings of readSettings  F:15046-15094.  Keys (double): tol, abs_tol, bc_tol,
 * ref_scale, bc_relax, mu_0, lambda_0 (the <ref> material), eps_g, eps_a (<laminate_mixing>).
 * Keys (integer): maxiter, mixing_rule (FG_MIXING_*), update_ref (0 = "never"), mode (0 = elasticity, 1 = heat /
 * porous F:15224-15228: fields "epsilon" / "sigma" have 3 components (gradient, flux), "u" one (the potential); mu of
 * fg_set_phase is the conductivity; E6 / S6 / out6 arrays carry 3 values followed by zeros; Voigt mixing, basic scheme;
 * 2 = viscosity F:15234-15239: dual Stokes scheme DeltaOperatorStaggered F:20422-20460, mu of fg_set_phase is the
 * fluidity constant of the XML, "epsilon" holds the fluid stress, "sigma" the shear rate; Voigt mixing, basic scheme),
 * gamma_scheme (0 = staggered, 1 = collocated: GammaOperatorCollocated F:20302-20310, Fourier-space 6x6 Gamma0),
 * method (0 = basic scheme, runBasic F:21716-21805; 1 = conjugate gradients, runCGElasticity
 * F:23153-23247, the reference's default), error_estimator (0 = epsilon F:14591-14637; 1 = residual F:14382-14405, method
 * cg only; 2 = sigma F:14514-14587; 3 = energy F:14410-14468; 4 = none F:14370-14378 -- 2 and 3 re-measure <sigma> / <W> of
 * the strain field after every iteration (two more sweeps), the conjugate gradients then iterate on the strain field), and the implementation switches u_loop (2 = default: the loop carries the
 * displacement, fast kernels; 1 = the same with the reference's operation order, iterates bit-identical to 0; 0 = the
 * strain field is the state, one kernel per reference routine where fuse_* are 0 too),
 * fuse_stress_div, fuse_x (1 = default; 0 selects the one-kernel-per-routine pipeline), u_tile (1 = default: the
 * LDS-tiled displacement sweep where the grid allows -- any non-zero value means "tiled", the tile shape follows from the
 * grid; 0 = the untiled sweep everywhere),
 * cg_fused (-1 = default: where the tiled sweep fits; 0 / 1: method = cg in displacement / potential space with the vector
 * work of an iteration as two tiled sweeps and the direction update inside the operator's sweep -- out of place, nine more
 * components; falls back to the four-kernel form when they do not fit; in the scalar modes (potential space) a registered
 * convergence callback also selects the four-kernel form, whose accessors see the iterate in place -- its residual history
 * equals the fused one's to rounding (different summation order), tests/test_gpu_cg_fused.py),
 * phi_sweep (1 = default: with two phases whose fractions are complementary bit for bit the tiled sweep reads phi_1 and
 * forms the effective moduli itself; 0 = always the two precomputed moduli arrays), laminate_overlap (1 = default: the interface kernels of the laminate correction run on a second stream beside the
 * displacement sweep; 0 = one stream), slab_split (slab driver: 1 = one all-to-all per component, overlapping the transforms of the next component; 0 = one
 * exchange for the three components; -1 = by slab size, default), slab_interleave (-1 = default: in the one-exchange mode a
 * peer's three components travel as ONE message where the sizes allow; 0 = one message per peer and component),
 * slab_loopback (test mode on one GPU: a lone slab that is
 * connected to a transport sends its all-to-all blocks and halo planes to itself through that transport),
 * x_layout (-1 = default: on grids whose three components exceed 1 GB the spectrum between the y passes and the fused x pass
 * is stored x-contiguous, [zc/8][y][x][8]; 0 / 1 force it off / on), plane_fft (-1 = default: where the complex z-y plane
 * fits the LDS -- ny * nz <= 128^2, 256 x 64 -- the z and y transforms of a plane run as ONE kernel; 0 = separate passes),
 * error_estimator (0 epsilon = default, 1 residual, 2 sigma, 3 energy, 4 none: create_error_estimator F:14940-14972),
 * pair_chunk (0 = default; P: the z and y transform passes in runs of P x planes), staged_copy (-1 = default: field downloads of
 * 8 MB and more through the pinned-buffer pipeline; 1 both directions, 0 one strided copy), stage_chunk_kb (pipeline stage),
 * joint_x (tile kernels of the lengths that are not powers of two: 1 = default, fused x pass on one joint LDS image of the three
 * components; 0 = one image per component), tile_plans (1 = default: tile kernels built for one plan each where the plan is in
 * the tables of fg_fft_smooth_plans.h; 0 = the class kernels for every plan; a process-wide switch). */
int fg_set_option_d(fg_solver* s, const char* key, double value);
int fg_set_option_i(fg_solver* s, const char* key, long value);

/* setBCProjector  F:20599-20665: symmetric 6x6 Voigt projector, row-major. */
int fg_set_bc_projector(fg_solver* s, const double* P36);
int fg_set_convergence_callback(fg_solver* s, fg_callback cb, void* user);
/* FG::cancel  F:25190-25193: makes a running fg_run_load_case fail at the next iteration. */
int fg_cancel(fg_solver* s);

/* setStrain + setStress + LSSolver::run  F:21247-21398 (one load step, method=basic,
 * gamma_scheme=staggered).  S may be NULL (= 0).  *failed receives the reference's
 * return value (1 = stopped on an error such as a NaN residual, F:21202-21208). */
int fg_run_load_case(fg_solver* s, const double* E6, const double* S6, int* failed);

/* runLoadsteppingSolver  F:21584-21685 with the <loadsteps> of the project (F:15095-15119): step i = first .. nparams-1
 * prescribes params[i] * (E6, S6) and starts from the strain field of the step before (zeroed once, F:21379); after
 * every step the load-step action runs (performLoadstepActions F:21435-21447): cb(user, i), non-zero stops the run,
 * which then reports failure like the reference.  fg_run_load_case is the standard list {0, 1} with first = 1
 * (F:21591).  Option loadstep_extrapolation_order (int, 0..7; F:14696, default 0): from the second step on a step starts
 * from the polynomial through the converged strain fields of the last order + 1 steps (extrapolateLoadstepPolynomial
 * F:21468-21514; single-GPU solvers).  Slab-decomposed solvers: collective, basic scheme and CG, no extrapolation. */
typedef int (*fg_loadstep_callback)(void* user, int istep);
int fg_run_load_steps(fg_solver* s, const double* E6, const double* S6, const double* params, int nparams, int first,
                      fg_loadstep_callback cb, void* user, int* failed);

/* n passes of basicScheme without convergence logic, and the same bracketed by HIP
 * events on the solver stream (benchmarks). */
int fg_iterate(fg_solver* s, const double* E6, int n);
int fg_time_iterations(fg_solver* s, const double* E6, int n, double* elapsed_ms);

/* results: getResiduals F:14766, iteration count, getSolveTime F:14767, calcMeanStress
 * F:17793-17811, calcMeanStrain (average F:10171), getVolumeFraction */
long fg_get_iterations(const fg_solver* s);
int fg_get_residuals(const fg_solver* s, double* out, int capacity); /* returns the count */
double fg_get_solve_time(const fg_solver* s);
int fg_mean_stress(fg_solver* s, double* out6);
int fg_mean_strain(fg_solver* s, double* out6);
int fg_volume_fraction(fg_solver* s, int p, double* out);

/* calcRefMaterial  F:22283-22313 and the resulting (mu_0, lambda_0) */
int fg_calc_ref_material(fg_solver* s);
int fg_get_ref_material(const fg_solver* s, double* mu_0, double* lambda_0);

/* get_raw_field  F:15396-15684.  Names: "epsilon" (6), "sigma" (6, evaluated with C0 = 0),
 * "u" (3), "phi" (nphases), "normals" (3); work buffers for stage tests: "tau" (6), "f" (3),
 * "f_hat" (3 complex components in the padded layout [nx][ny][nz/2+1][2]), "sumsq" (6 scalars).
 * The work buffers hold what the LAST fg_run_stage left there and nothing else: between stages "tau" is scratch of the
 * transform chain (the x-contiguous spectrum layout of large grids, the all-to-all halves of a slab solver, CG vectors),
 * so after fg_iterate / fg_run_* its contents are unspecified -- read "tau" directly after FG_STAGE_STRESS only.  The
 * same holds for fg_device_pointer("tau" | "f", c). */
int fg_field_components(const fg_solver* s, const char* name);
int fg_get_field(fg_solver* s, const char* name, double* out);
int fg_set_field(fg_solver* s, const char* name, const double* in);

/* Device-side access for zero-copy callers (slab-decomposed driver): pointer to padded
 * component `comp` ([nx][ny][2*(nz/2+1)] doubles) and the solver's hipStream_t. */
void* fg_device_pointer(fg_solver* s, const char* name, int comp);
void* fg_get_stream(fg_solver* s);
int fg_synchronize(fg_solver* s);

/* Single stages on the solver's buffers (parity tests and profiling).  For FG_STAGE_G0,
 * E6[0] carries alpha (default -1).  With timing enabled every kernel of a pass is bracketed
 * by HIP events on the solver stream; fg_get_stage_times reports the accumulated milliseconds
 * of the FG_NUM_TIMED_KERNELS kernels in launch order: stress, div, r2c_z, c2c_y_fwd,
 * c2c_x_fwd, g0, c2c_x_inv, c2c_y_inv, c2r_z, eps_norm, and the number of passes timed.  Enabling (from disabled)
 * resets the accumulators. */
#define FG_NUM_TIMED_KERNELS 10
int fg_run_stage(fg_solver* s, int stage, const double* E6);
int fg_enable_stage_timing(fg_solver* s, int enable);
int fg_get_stage_times(const fg_solver* s, double* ms /* [FG_NUM_TIMED_KERNELS] */, long* count);
/* What a pair of HIP events reads with nothing between them on this solver's stream (measured when timing is switched
 * on, milliseconds); it has been subtracted from every figure fg_get_stage_times reports. */
int fg_get_stage_timing_bias(const fg_solver* s, double* ms);
/* Sizes and set-up facts of this solver (no counterpart in the reference; bench.py prices the laminate correction with them):
 * "interface_voxels" / "affected_voxels" = lengths of the laminate correction's voxel lists (0 until a laminate pass built
 * them); "pair_chunk_planes" = x planes per chunk of the paired z / y transform passes (option pair_chunk; 0 = whole-field
 * passes: stage timing then reports each pair in the slot of its first pass).  Unknown names give -1. */
long fg_get_counter(const fg_solver* s, const char* name);

/* Measurement helper (no counterpart in the reference): achieved HBM bandwidth of a streaming copy a = b and of the
 * triad a = b + s*c on `device`, arrays of `megabytes` MB each, best of `reps` launches, in GB/s of bytes moved
 * (2 resp. 3 arrays).  The "measured roofline" that bench.py reports beside the 8 TB/s peak. */
int fg_hbm_stream(int device, int megabytes, int reps, double* copy_GBps, double* triad_GBps);

/* ---- slab decomposition over the GPUs of a node ----------------------------
 * The reference is single-process; this is the multi-GPU counterpart of one LSSolver.  Rank r
 * of nranks owns the x-planes [r*nx/nranks, (r+1)*nx/nranks) of every field (nx and ny must be
 * divisible by nranks); all arrays passed to / returned by the calls above then have the LOCAL
 * shape [ncomp][nx/nranks][ny][nz]. */
fg_solver* fg_create_slab(int nx, int ny, int nz, double dx, double dy, double dz, int device, int rank, int nranks);

/* The loop of a slab solver runs inside the library: the same entry points as for one GPU (fg_run_load_case,
 * fg_iterate, fg_time_iterations, fg_mean_stress, fg_mean_strain, fg_volume_fraction, fg_calc_ref_material,
 * fg_get_field with LOCAL shapes) become collective calls once the solver is connected to a transport.  Per pass
 * (basicScheme F:20558-20578 with GammaOperatorStaggered F:20288-20300 cut along x):
 *
 *   displacement sweep on the x-slab (u with +-1 halo planes -> norms of eps_k, f_{k+1})   | all-reduce of 6..12 doubles
 *   per component c: z r2c, y c2c writing the all-to-all layout                             | all-to-all(c)  (overlaps c+1)
 *   x c2c + Green operator + x c2c^-1 on the y-slab (three components, one kernel)          | all-to-all(c) back
 *   per component c: y c2c^-1 reading the all-to-all layout, z c2r                          | halo planes of u
 *
 * all stream-ordered: exchanges run on a second stream joined by events, the host waits only for the norms of the
 * stop rule.  Grids the tiled sweep does not fit take the strain-state pipeline (polarisation, halo of tau, divergence,
 * the same transform chain, halo of u, strain + norms) under the same driver.  method = cg (displacement space where the
 * tiled sweep applies, strain space otherwise), fg_run_load_steps, mode = heat / porous (potential sweep with halo planes, any
 * grid; basic scheme also with mixed boundary conditions, CG with prescribed mean gradients) and mode = viscosity (strain-state
 * pipeline, basic scheme and CG) run on slabs as well.
 *
 * Transports (exactly one per solver, before the first collective call):
 *   fg_slab_connect_rccl      one process per GPU; RCCL (ncclSend / ncclRecv groups, ncclAllReduce) over xGMI.
 *                             Rank 0 obtains the id with fg_comm_unique_id and hands it to all ranks out of band.
 *   fg_slab_group_create      all nranks slabs in THIS process on ONE device and one stream (single-GPU tests,
 *                             bench.py --force-slab); a collective call on any member drives all members.
 *   fg_slab_connect_callback  the caller moves the bytes (multi-process tests over gloo). */
#define FG_COMM_ID_BYTES 128
int fg_comm_unique_id(char* id /* [FG_COMM_ID_BYTES] */);
int fg_slab_connect_rccl(fg_solver* s, const char* id /* [FG_COMM_ID_BYTES] */);
int fg_slab_group_create(int nx, int ny, int nz, double dx, double dy, double dz, int device, int nranks,
                         fg_solver** out /* [nranks] */);
typedef struct fg_xop {
  int send;             /* 1 = send, 0 = receive */
  int peer;
  void* ptr;            /* device pointer */
  unsigned long bytes;
} fg_xop;
/* exchange: the solver's streams are drained; sends / receives between two ranks match in list order.
 * allreduce: in place on `n` host values (sum, or element-wise minimum when min_op != 0).  Return 0 on success. */
typedef int (*fg_exchange_fn)(void* user, const fg_xop* ops, int nops);
typedef int (*fg_allreduce_fn)(void* user, double* values, int n, int min_op);
int fg_slab_connect_callback(fg_solver* s, fg_exchange_fn exchange, fg_allreduce_fn allreduce, void* user);
/* name of the connected transport ("rccl", "local", "callback") or "" */
const char* fg_slab_transport(const fg_solver* s);

/* Stop and error decisions of a collective run are themselves collective: every reduction of the loop carries a flag word
 * (device error flag of any rank, fg_cancel on any rank), and fg_run_load_case returns / fails on every rank in the same
 * pass.  fg_cancel from another thread takes effect with the next pass's reduction.  Convergence callbacks (F:21215) may be
 * installed on some ranks only and may answer differently: when any rank has one, every pass ends with a vote and the
 * run stops everywhere as soon as one callback asks for it (the transform chain of the next pass is then no longer
 * enqueued speculatively, which costs the overlap of the host with the device).  The same holds for fg_run_load_steps: the
 * outcome of every step and the answer of the load-step callback (which may exist on one rank only) are summed over the
 * ranks after the step, and every rank returns in the same step when any rank failed or asked to stop. */

/* Measurement: with stage timing on (fg_enable_stage_timing) the exchanges of a slab solver are bracketed by HIP events on
 * the exchange stream and waited for one by one; ms[0] all-to-all forward, ms[1] all-to-all backward, ms[2] halo planes,
 * ms[3] all-reduces, accumulated over the passes fg_get_stage_times counts. */
int fg_get_comm_times(const fg_solver* s, double* ms /* [4] */);
/* PCI bus id ("0000:05:00.0") of HIP device `device`: lets a launcher show that N ranks sit on N different GPUs. */
int fg_device_pci_bus_id(int device, char* out, int capacity);

/* The exchange plan (pure index arithmetic, no GPU): ops of exchange `what` for `rank`, offsets / counts in doubles
 * relative to the named buffer.  Returns the number of ops (<= capacity) or -1.  self_copy[2] (may be NULL) receives
 * the rank's own all-to-all block as {source, destination}. */
#define FG_PLAN_A2A_FORWARD 0   /* x-slabs -> y-slabs: component `comp` = 0..2 (blocks of one component), or comp = -1: ONE
                                 * message per peer holding its three components ([P][3][nx/P][ny/P][nzp] on both sides) */
#define FG_PLAN_A2A_BACKWARD 1
#define FG_PLAN_HALO_U 2        /* +-1 planes of the three displacement components */
#define FG_PLAN_HALO_MODULI 3   /* the same for the two effective-moduli arrays (once per geometry) */
#define FG_PLAN_HALO_TAU 4      /* strain-state pipeline: tau0 | tau5, tau4 */
#define FG_BUF_SPECTRUM_X 0     /* [3][P][nx/P][ny/P][nzp] blocked x-slab spectrum */
#define FG_BUF_SPECTRUM_Y 1     /* [3][nx][ny/P][nzp] y-slab spectrum (shares its memory with the divergence field f) */
#define FG_BUF_U 2              /* [3][nx/P + 4][ny][nzp] displacement with spare planes */
#define FG_BUF_MODULI 3         /* [2][nx/P + 4][ny][nzp] */
#define FG_BUF_HALO_SEND_LO 4
#define FG_BUF_HALO_SEND_HI 5
#define FG_BUF_HALO_RECV_LO 6
#define FG_BUF_HALO_RECV_HI 7
typedef struct fg_plan_op {
  int send;
  int peer;
  int buffer;
  long offset;
  long count;
} fg_plan_op;
int fg_slab_plan(int nx, int ny, int nz, int nranks, int rank, int what, int comp, fg_plan_op* ops, int capacity,
                 fg_plan_op* self_copy);

/* ---- geometry pre-processing (kernels on HIP device `device`) ------------------------
 * Analytic shapes placed with <place_fiber> (F:25788-25822) -> phase volume fractions and
 * interface normals: LSSolver::initPhi F:17489-17581 (adaptive sub-voxel integration,
 * integratePhiVoxel F:16622-16752, plane cuts F:1385-1577) and the NORMALS sampling
 * F:6905-6925.  kind 0 = capsule (centre c, axis a, total length L, radius R; L = 0 is a
 * sphere), kind 1 = half space (point c, outward normal a).  phi is [nphases][nx][ny][nz]
 * BEFORE normalizePhi (the matrix phase is all ones); normals (may be NULL) is
 * [3][nx][ny][nz]; real_volume (may be NULL) receives the analytic volume per phase. */
typedef struct fg_fiber {
  int kind;
  int material;
  double c[3];
  double a[3];
  double L;
  double R;
} fg_fiber;

int fg_voxelize(const fg_fiber* fibers, int nfibers, int nx, int ny, int nz, double dx, double dy, double dz,
                const double* x0, int nphases, int matrix_mat, int smooth_levels, double smooth_tol,
                double* phi, double* normals, double* real_volume, int device, char* err, int errlen);

/* Test hook of the voxeliser (no counterpart in the reference): threads per interface voxel = 8^depth for the calls that
 * follow in this process, depth 0 ... 3; -1 (the default) = chosen from the number of interface voxels.  Every depth gives
 * bit-identical fractions (tests/test_gpu_voxelize.py).  Returns the previous setting. */
int fg_voxelize_team_depth(int depth);

#ifdef __cplusplus
}
#endif

#endif /* FIBERGEN_AMD_H */
