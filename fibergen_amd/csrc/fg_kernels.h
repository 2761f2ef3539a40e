// Host-callable launchers of the streaming kernels (fg_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "fg_stage_math.h"

namespace fg {

template <int N>
struct FieldPtrs {
  double* p[N];
};

struct Vec6 {
  double v[6];
};

// Separable factors of the staggered Green operator (F:19856-19876), device arrays:
// kpm[a][m] = sin(xi)/h, kp[a][m] = kpm e^{i xi}; the z tables hold nzc entries.
struct G0Tables {
  const double* kpm[3];
  const cplx* kp[3];
};

// x-halo planes ([ny][nzp] doubles each) of an x-slab: lo = plane just below the slab, hi = just above.
// Null pointers mean "periodic wrap inside this field" (single-GPU case).
//   div : lo[0] = tau0 ;            hi[0] = tau5, hi[1] = tau4
//   eps : lo[0] = u1, lo[1] = u2 ;  hi[0] = u0
struct XHalo {
  const double* lo[2];
  const double* hi[2];
};

// collocated scheme: xi_a[m] = m_signed / d_a per axis (F:19385, 19411-19424), the z table holds nzc entries
struct XiTables {
  const double* xi[3];
};

struct G0Layout {
  int transposed;  // 0: [nx][ny][nzc]   1: y-slab [nyl][nx][nzc]
  int nyl, jj0;
};

// Scalar modes (heat / porous): conductivities of the phases, calcStress factors (beta = -alpha 2 mu0)
struct ScalarParams {
  int n;
  double mu[kMaxPhases];
  double alpha, beta;
};

constexpr int kMaxReduceBlocks = 4096;  // partial-sum rows of the two-stage reductions

int reduce_blocks(const Grid& g);
long partial_rows(const Grid& g);  // rows (of up to 8 doubles) the partial-sum buffer must hold

void launch_stress(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps, const FieldPtrs<kMaxPhases>& phi,
                   const FieldPtrs<3>& normals, const FieldPtrs<6>& tau, int* error_flag, hipStream_t s);
// meanW  F:12239-12262: the sum of the energy densities (sp.alpha = 1; the caller divides by the number of voxels) in out6[0]
void launch_energy_mean(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps, const FieldPtrs<kMaxPhases>& phi,
                        const FieldPtrs<3>& normals, double* partial, double* out6, int* error_flag, hipStream_t s);
void launch_stress_mean(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps,
                        const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& normals, double* partial, double* out6,
                        int* error_flag, hipStream_t s);
void launch_stress_const(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<6>& eps, const FieldPtrs<6>& tau,
                         hipStream_t s);
void launch_stress_div_voigt(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps,
                            const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& f, hipStream_t s);
void launch_u_stress_div_voigt(const Grid& g, const StressParams& sp, const FieldPtrs<3>& u,
                               const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& f, const Vec6& E, double* partial,
                               double* sumsq6, hipStream_t s);
void launch_effective_moduli(const Grid& g, const PhaseTable& pt, const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<2>& mod,
                             hipStream_t s);
void launch_u_fast(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& u, const FieldPtrs<2>& mod,
                   const FieldPtrs<3>& f, const Vec6& E, double* partial, double* sumsq6, hipStream_t s);
// tiled variant (every strain / polarisation value computed once; y neighbours through LDS, x by marching)
bool u_tile_supported(const Grid& g);
// the polarisation + divergence half of the tiled sweep for a stored strain field (same grids as u_tile_supported):
// f = div((C - C0) : eps) with the effective moduli of k_effective_moduli, sum6 = sums of the polarisation components
void launch_eps_tile(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<6>& eps, const FieldPtrs<2>& mod,
                     const FieldPtrs<3>& f, double* partial, double* sum6, hipStream_t s);
// two_phase != nullptr: mod.p[0] is phi_1 of two complementary phases (launch_complement_check), the sweep forms the moduli
void launch_u_tile(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& u, const FieldPtrs<2>& mod,
                   const FieldPtrs<3>& f, const Vec6& E, double* partial, double* sumsq6, hipStream_t s, bool sum_tau = false,
                   const PhaseTable* two_phase = nullptr);
// the tiled sweep on the NEW search direction of the conjugate gradients, formed on the fly: p_new = r + b p_old with
// b = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small), stored to p_new (a buffer of its own), f = div((C - C0) : grad_s p_new)
void launch_u_tile_cg(const Grid& g, double mu_0, double lambda_0, const FieldPtrs<3>& p_old, const FieldPtrs<3>& r,
                      const FieldPtrs<3>& p_new, const FieldPtrs<2>& mod, const FieldPtrs<3>& f, const Vec6& E, const double* sc,
                      int i_num, int i_den, double nvox, double small, double* partial, double* sumsq6, hipStream_t s,
                      const PhaseTable* two_phase);
void launch_complement_check(const Grid& g, const double* phi0, const double* phi1, int* flag, hipStream_t s);
// u_k -> sums of squares of eps_k and tau = (C - C0) : eps_k for any mixing rule (strain never stored)
void launch_u_stress(const Grid& g, const StressParams& sp, const FieldPtrs<3>& u, const FieldPtrs<kMaxPhases>& phi,
                     const FieldPtrs<3>& normals, const FieldPtrs<6>& tau, const Vec6& E, double* partial, double* sumsq6,
                     int* error_flag, hipStream_t s);
// CG in displacement space: inner products of staggered gradients (mode 0: p:(p-w) -> out[0]; mode 1: sums of squares of
// E + grad_s a -> out[0..5] and grad_s b : grad_s b -> out[6]) and the point-wise vector updates
void launch_cgu_dot(int mode, const Grid& g, const FieldPtrs<3>& a, const FieldPtrs<3>& b, const Vec6& E, double* partial,
                    double* out7, hipStream_t s);
void launch_cgu_axpy(int mode, const Grid& g, const FieldPtrs<3>& x, const FieldPtrs<3>& y, const FieldPtrs<3>& r,
                     const FieldPtrs<3>& w, const double* sc, int i_num, int i_den, double nvox, double small, hipStream_t s,
                     long count = 0 /* doubles per component to update; 0 = g.n (x-slabs: + the spare planes) */);
// out-of-place point-wise updates on `count` doubles per component from offset `off` (the spare planes of an x-slab's vectors
// when the fused sweeps below wrote the own planes): mode 0: xo = x + a y, ro = r - a (y - w); mode 1: xo = r + a y
void launch_cgu_axpy_oop(int mode, const FieldPtrs<3>& x, const FieldPtrs<3>& y, const FieldPtrs<3>& r, const FieldPtrs<3>& w,
                         const FieldPtrs<3>& xo, const FieldPtrs<3>& ro, const double* sc, int i_num, int i_den, double nvox,
                         double small, long off, long count, hipStream_t s);
// the same sweeps in k_u_tile's tiling (grids of u_tile_supported), the update fused with its norms and OUT OF PLACE:
// mode 0: out7[0] = grad_s a : (grad_s a - grad_s b); mode 1: A = a + alpha y -> ao, B = b - alpha (y - w) -> bo with
// alpha = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small), out7[0..5] = sums of (E + grad_s A)_c^2, out7[6] = B : B
void launch_cgu_tile(int mode, const Grid& g, const FieldPtrs<3>& a, const FieldPtrs<3>& b, const FieldPtrs<3>& y,
                     const FieldPtrs<3>& w, const FieldPtrs<3>& ao, const FieldPtrs<3>& bo, const Vec6& E, const double* sc,
                     int i_num, int i_den, double nvox, double small, double* partial, double* out7, hipStream_t s);
// interface voxels (some phase fraction strictly between 0 and 1): allocates and fills the list of their element
// offsets in voxel order (*list, hipFree by the caller), returns the count
unsigned launch_mixed_list(const Grid& g, int nph, const FieldPtrs<kMaxPhases>& phi, unsigned** list, hipStream_t s);
// laminate mixing as a correction of the Voigt sweep (see k_interface_strain): the voxels whose divergence stencil touches
// an interface voxel, in voxel order, with 8 slots each (*aff, *slots allocated here); per pass the polarisation
// difference at the interface voxels [n][6] and its divergence added to f
unsigned launch_affected_list(const Grid& g, const unsigned* list, unsigned n, unsigned** aff, int** slots, hipStream_t s);
// laminate correction in compact form (see k_interface_strain): static copies of the phase fractions [nph][n] and normals
// [3][n] of the interface voxels (once per geometry), and per pass eps_j -> d_j (epsc: scratch [6][n], dtau: [n][6])
void launch_interface_static(const Grid& g, int nph, const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& normals,
                             const unsigned* list, unsigned n, double* phic, double* nrmc, hipStream_t s);
void launch_interface_delta(const Grid& g, const StressParams& sp, const FieldPtrs<3>& u, const Vec6& E, const unsigned* list,
                            unsigned n, double* epsc, const double* phic, const double* nrmc, double* dtau, int* error_flag,
                            hipStream_t s);
void launch_sum_dtau(const double* dtau, unsigned n, double* partial, double* out6, hipStream_t s);
void launch_delta_div(const Grid& g, const unsigned* aff, const int* slots, unsigned n, const double* dtau,
                      const FieldPtrs<3>& f, hipStream_t s);
// x-slabs: the difference field of the boundary planes as dense planes for the neighbours (lo2: d5, d4 of the first plane;
// hi1: d0 of the last plane) and the neighbours' planes applied to f (from_lo1: d0 of plane -1; from_hi2: d5, d4 of plane nx)
void launch_delta_pack(const Grid& g, const unsigned* list, unsigned n, const double* dtau, double* lo2, double* hi1, hipStream_t s);
void launch_delta_div_halo(const Grid& g, const double* from_lo1, const double* from_hi2, const FieldPtrs<3>& f, hipStream_t s);
void launch_add_small(double* out, const double* in, int n, hipStream_t s);
void launch_div(const Grid& g, const FieldPtrs<6>& tau, const FieldPtrs<3>& f, const XHalo& h, hipStream_t s);
void launch_g0(const Grid& g, const FieldPtrs<3>& fh, const G0Tables& tb, double c10, double c20, const G0Layout& lay,
               hipStream_t s);
void launch_gamma_collocated(const Grid& g, const FieldPtrs<6>& th, const XiTables& xt, double c10, double c20, double beta,
                             const Vec6& E, hipStream_t s);
void launch_eps_norm(const Grid& g, const FieldPtrs<3>& u, const FieldPtrs<6>& eps, const Vec6& E, const Vec6& R,
                     bool add_R, double* partial, double* sumsq6, const XHalo& h, hipStream_t s);
// viscosity: eta = (E - coef tau_sum / nvox) + sym grad u + coef tau, sums of squares (tau_sum on the device)
// viscosity mode without a stored polarisation: divergence of the polarisation with its six sums (<tau>), and the
// Delta-operator tail re-evaluating the polarisation from the strain the pass started from
void launch_stress_div_sum_voigt(const Grid& g, const StressParams& sp, const FieldPtrs<6>& eps,
                                const FieldPtrs<kMaxPhases>& phi, const FieldPtrs<3>& f, double* partial, double* sum6,
                                hipStream_t s);
void launch_eps_delta_recompute(const Grid& g, const FieldPtrs<3>& u, const FieldPtrs<6>& eps_old, const StressParams& sp,
                                const FieldPtrs<kMaxPhases>& phi, const double* tau_sum, double nvox, const Vec6& E,
                                double coef, const FieldPtrs<6>& eps, double* partial, double* sumsq6, hipStream_t s);
void launch_eps_delta(const Grid& g, const FieldPtrs<3>& u, const FieldPtrs<6>& tau, const double* tau_sum, double nvox,
                      const Vec6& E, double coef, const FieldPtrs<6>& eps, double* partial, double* sumsq6, hipStream_t s);
void launch_copy(const double* src, double* dst, long ndoubles, hipStream_t s);
// out = sum_{i<n} w[i] in[i] (n <= 8)
void launch_lincomb(int n, const double* const* in, const double* w, double* out, long ndoubles, hipStream_t s);
void launch_cg(int mode, const Grid& g, const FieldPtrs<6>& x, const FieldPtrs<6>& y, const FieldPtrs<6>& z, const Vec6& E,
               double a, double* partial, double* out6, hipStream_t s);
// strain-space CG with the scalars on the device: mode 5: eps += a p, r -= a (p - w) in one sweep, out7 = sums of eps_c^2 and r:r;
// mode 6: p = r + a p;  a = (sc[i_num] / nvox + small) / (sc[i_den] / nvox + small)
void launch_cg_dev(int mode, const Grid& g, const FieldPtrs<6>& e, const FieldPtrs<6>& r, const FieldPtrs<6>& p, const FieldPtrs<6>& w,
                   const double* sc, int i_num, int i_den, double nvox, double small, double* partial, double* out7, hipStream_t s);
void launch_set_const6(const Grid& g, const FieldPtrs<6>& x, const Vec6& E, hipStream_t s);
void launch_sum6(const Grid& g, const FieldPtrs<6>& x, bool square, double* partial, double* out6, hipStream_t s);
void launch_sum1(const Grid& g, const double* x, double* partial, double* out1, hipStream_t s);
// scalar modes (fg_kernels_scalar.hip)
void launch_sc_sweep(const Grid& g, const ScalarParams& sp, const double* T, const FieldPtrs<kMaxPhases>& phi, double* f,
                     const Vec6& E, double* partial, double* sumsq6, hipStream_t s);
// fast variant: a = per-voxel effective conductivity (launch_effective_moduli with 2 mu_p := mu_p, first array)
// sumtau3 != nullptr: also the three sums of the flux polarisation (mixed boundary conditions); returns whether they were
// produced (the LDS-tiled form does, the untiled one does not)
bool sc_sweep_tiled(const Grid& g);   // launch_sc_sweep_fast takes the tiled kernel (the one that can carry the sums of tau)
bool launch_sc_sweep_fast(const Grid& g, double mu_0, const double* T, const double* a, double* f, const Vec6& E,
                          double* partial, double* sumsq6, hipStream_t s, double* sumtau3 = nullptr);
void launch_sc_cg_axpy_oop(int mode, const double* x, const double* y, const double* r, const double* w, double* xo, double* ro,
                           const double* sc, int i_num, int i_den, double nvox, double small, long off, long count, hipStream_t s);
// CG in potential space, tiled and fused like launch_cgu_tile / launch_u_tile_cg (grids of sc_sweep_tiled)
void launch_sc_cgu_tile(int mode, const Grid& g, const double* a, const double* b, const double* y, const double* w, double* ao,
                        double* bo, const Vec6& E, const double* sc, int i_num, int i_den, double nvox, double small, double* partial,
                        double* out7, hipStream_t s);
void launch_sc_sweep_cg(const Grid& g, double mu_0, const double* p_old, const double* r, double* p_new, const double* a, double* f,
                        const Vec6& E, const double* sc, int i_num, int i_den, double nvox, double small, double* partial,
                        double* sumsq6, hipStream_t s);
void launch_sc_grad(const Grid& g, const double* T, const FieldPtrs<3>& out, const Vec6& E, double* partial,
                    double* sumsq6, hipStream_t s);
void launch_sc_flux(const Grid& g, const ScalarParams& sp, const FieldPtrs<3>& gr, const FieldPtrs<kMaxPhases>& phi,
                    const FieldPtrs<3>& out, hipStream_t s);
void launch_sc_flux_mean(const Grid& g, const ScalarParams& sp, const FieldPtrs<3>& gr, const FieldPtrs<kMaxPhases>& phi,
                         double* partial, double* out6, hipStream_t s);
void launch_sc_div(const Grid& g, const FieldPtrs<3>& x, double scale, double* y, hipStream_t s);
void launch_g0_heat(const Grid& g, double* fh, const G0Tables& tb, double c10, hipStream_t s, int jj0 = 0);
// CG in potential space (scalar modes): dot products of forward-difference gradients, point-wise updates
void launch_sc_cg_dot(int mode, const Grid& g, const double* a, const double* b, const Vec6& E, double* partial, double* out7,
                      hipStream_t s);
void launch_sc_cg_axpy(int mode, const Grid& g, double* x, double* y, double* r, const double* w, double a, hipStream_t s,
                       long count = 0 /* doubles to update; 0 = g.n (x-slabs: + the spare planes) */);
void launch_sc_minmax(const Grid& g, const ScalarParams& sp, const FieldPtrs<kMaxPhases>& phi, double* partial,
                      double* out2, hipStream_t s);
void launch_tangent_minmax(const Grid& g, const PhaseTable& pt, int mixing, const FieldPtrs<kMaxPhases>& phi,
                           double* partial, double* out2, int* error_flag, hipStream_t s);

}  // namespace fg
