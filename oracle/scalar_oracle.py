"""NumPy restatement of fibergen's basic scheme in the scalar modes (mode=heat / mode=porous).

TEST INFRASTRUCTURE ONLY -- the checker, never the product path (see oracle/ls_oracle.py).

The scalar modes (BASELINE config 5) solve  div((kappa - kappa0) g) ... for the gradient
field g = E + grad T of a potential T (temperature / pressure): fields have 3 components, the potential
one.  ``F`` = /root/reference/src/fibergen.cpp.  Each method cites the routine it restates.

Pinning (tests/test_oracle_pins.py::test_scalar_*): the reference holds no numeric fixtures for these modes;
pinned against closed forms that the reference's discretisation reproduces exactly -- layered media
(series = harmonic mean across the layers, parallel = arithmetic mean along them), the homogeneous medium,
the Wiener and Hashin-Shtrikman bounds -- and the operator identity  grad(G0(div(2 mu0 grad T))) = grad T.
"""
from __future__ import annotations

import math

import numpy as np

from .ls_oracle import EPS, SMALLEST, LSOracle

VOIGT_THRESHOLD = 10 * EPS


class ScalarOracle:
    """LSSolver<double,double,3> with mode=heat|porous, method=basic, gamma_scheme=staggered,
    mixing_rule=voigt, law="iso" (ScalarLinearIsotropicMaterialLaw F:11158-11215: flux = mu * gradient)."""

    def __init__(self, nx, ny, nz, mus, phis, dx=1.0, dy=1.0, dz=1.0, tol=1e-4, abs_tol=EPS, bc_tol=1e-3,
                 maxiter=10000, ref_scale=1.0, mu_0=float("nan"), update_ref="loadstep"):
        self.nx, self.ny, self.nz = nx, ny, nz
        self.dx, self.dy, self.dz = dx, dy, dz
        self.mus = list(mus)
        self.phis = [np.asarray(p, dtype=np.float64) for p in phis]
        self.tol, self.abs_tol, self.bc_tol, self.maxiter = tol, abs_tol, bc_tol, maxiter
        self.ref_scale, self.mu_0, self.update_ref = ref_scale, mu_0, update_ref
        self.N = nx * ny * nz
        self.eps = np.zeros((3, nx, ny, nz))
        self.residuals = []
        self.callback = None
        self.error = None
        self.bc_relax = 1.0
        self.lambda_0 = 0.0
        self.set_bc_projector(np.eye(3))
        # Green-operator tables are those of the elasticity path (same k+ / k- factors, F:19766-19815)
        self._tables = LSOracle(nx, ny, nz, dx, dy, dz).g0_axis_tables()

    # -- constitutive ------------------------------------------------------------------------
    def pk1(self, g, alpha=1.0):
        """VoigtMixedMaterialLaw<.,.,3>::PK1  F:12752-12761 over ScalarLinearIsotropicMaterialLaw::PK1
        F:11182-11198: S_m (+)= E_m * ((phi*alpha)*mu), phases with phi <= 10 eps skipped."""
        P = np.zeros_like(g)
        first = np.ones(g.shape[1:], dtype=bool)
        for phi, mu in zip(self.phis, self.mus):
            live = phi > VOIGT_THRESHOLD
            alpha_mu = (phi * alpha) * mu
            for m in range(3):
                term = g[m] * alpha_mu
                P[m] = np.where(live, np.where(first, term, P[m] + term), P[m])
            first &= ~live
        return P

    def calc_stress(self, mu_0, g, alpha=1.0):
        """calcStress  F:18134-18184 with dim 3: tau = P(g) + beta g, beta = -alpha 2 mu0 (lambda0 = 0)."""
        beta = -alpha * 2 * mu_0
        P = self.pk1(g, alpha)
        if beta != 0:
            P = P + beta * g
        return P

    def mean_stress(self, g=None):
        """calcMeanStress -> meanPK1  F:17793-17811, F:12312-12351"""
        g = self.eps if g is None else g
        return self.pk1(g, 1.0 / self.N).reshape(3, -1).sum(axis=1)

    def mean_strain(self):
        return self.eps.reshape(3, -1).sum(axis=1) / self.N

    # -- difference operators ----------------------------------------------------------------
    def _h(self):
        return self.nx / self.dx, self.ny / self.dy, self.nz / self.dz

    def div_heat(self, x):
        """divOperatorStaggeredHeat  F:18914-18975: backward differences, accumulated x, y, z."""
        hx, hy, hz = self._h()
        y = (x[0] - np.roll(x[0], 1, axis=0)) * hx
        y = y + (x[1] - np.roll(x[1], 1, axis=1)) * hy
        y = y + (x[2] - np.roll(x[2], 1, axis=2)) * hz
        return y

    def eps_heat(self, E, T):
        """epsOperatorStaggeredHeat  F:18697-18760: forward differences of the potential plus E."""
        hx, hy, hz = self._h()
        y = np.empty((3,) + T.shape)
        y[0] = E[0] + (np.roll(T, -1, axis=0) - T) * hx
        y[1] = E[1] + (np.roll(T, -1, axis=1) - T) * hy
        y[2] = E[2] + (np.roll(T, -1, axis=2) - T) * hz
        return y

    # -- Green operator ------------------------------------------------------------------------
    def g0_heat(self, mu_0, f, alpha=1.0):
        """G0OperatorStaggeredHeat  F:20118-20135: fftVector(.,1) (1/N forward, F:18481-18510),
        G0OperatorFourierStaggeredHeat  F:19759-19823 (c1 = c10/|k|^2, c10 = -alpha/(2 mu0), zero mode 0),
        fftInvVector."""
        fh = np.fft.rfftn(f, axes=(0, 1, 2)) * (1 / float(self.N))
        c10 = -alpha / (2 * mu_0)
        (s0, _), (s1, _), (s2, _) = self._tables
        norm2 = (s0 * s0)[:, None, None] + (s1 * s1)[None, :, None] + (s2 * s2)[None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            th = (c10 / norm2) * fh
        th[0, 0, 0] = 0
        return np.fft.irfftn(th, s=(self.nx, self.ny, self.nz), axes=(0, 1, 2)) * float(self.N)

    def potential(self):
        """get_raw_field("u") in the scalar modes  F:15536-15541: T = G0(div(C0 : g)) with alpha = 1."""
        sigma = 2 * self.mu_0 * self.eps       # calcStressConst with lambda0 = 0  F:17973-18020
        return self.g0_heat(self.mu_0, self.div_heat(sigma), 1.0)

    # -- reference material --------------------------------------------------------------------------
    def calc_ref_material(self):
        """calcRefMaterial  F:22283-22313 over getRefMaterial / eig  F:12153-12236, F:12472-12559: the
        3x3 tangent of the Voigt-mixed scalar law is (sum_p phi_p mu_p) I."""
        mu_bar = np.zeros((self.nx, self.ny, self.nz))
        first = np.ones(mu_bar.shape, dtype=bool)
        for phi, mu in zip(self.phis, self.mus):
            live = phi > VOIGT_THRESHOLD
            t = (phi * 1.0) * mu
            mu_bar = np.where(live, np.where(first, t, mu_bar + t), mu_bar)
            first &= ~live
        lo, hi = float(mu_bar.min()), float(mu_bar.max())
        if lo < 0:
            lo = 0.0
        self.mu_0 = 0.5 * (lo + hi) * (0.5 * self.ref_scale)

    # -- boundary-condition projector (dim 3: plain 3x3 algebra, Voigt::Id4(3) = Id, F:501-512) -------------------------
    def set_bc_projector(self, P):
        """setBCProjector  F:20599-20665 for dim = 3: Q = Id - P, Q C0, M = (Q C0 Q)^+ by SVD, M Q;
        C0 = 2 mu0 Id + lambda0 II (F:20619)."""
        P = np.asarray(P, dtype=np.float64)
        se = math.sqrt(EPS)
        if P.shape != (3, 3) or np.linalg.norm(P - P.T) > se:
            raise RuntimeError("Projector is not symmetric")
        if np.linalg.norm(P - P @ P) > se:
            raise RuntimeError("Specified Projector is not a projector")
        self.BC_P = P
        self.BC_Q = np.eye(3) - P
        self._bc_matrices()

    def _bc_matrices(self):
        Q = self.BC_Q
        if not np.any(Q) or math.isnan(self.mu_0):
            self.BC_QC0 = np.zeros((3, 3)) if not np.any(Q) else Q * float("nan")
            self.BC_M = np.zeros((3, 3)) if not np.any(Q) else np.full((3, 3), float("nan"))
            self.BC_MQ = self.BC_M.copy()
            return
        C0 = 2 * self.mu_0 * np.eye(3) + self.lambda_0 * np.ones((3, 3))
        self.BC_QC0 = Q @ C0
        U, sv, VT = np.linalg.svd(self.BC_QC0 @ Q)
        thr = math.sqrt(EPS) * float(np.linalg.norm(sv))
        sinv = np.array([1.0 / x if abs(x) > thr else 0.0 for x in sv])
        self.BC_M = (VT.T * sinv) @ U.T
        self.BC_MQ = self.BC_M @ Q

    def calc_bc_mean(self, E, S):
        """calcBCMean  F:20242-20245"""
        return E + self.bc_relax * (self.BC_M @ (S - self.BC_QC0 @ E))

    # -- iteration ---------------------------------------------------------------------------------------
    def basic_scheme(self, E, g):
        """basicScheme  F:20558-20578 -> GammaOperatorStaggeredHeat  F:20342-20351 (alpha = -1):
        g <- E + grad G0(div((C - C0) g)) + R with initBCProjector / applyBCProjector  F:20228-20270."""
        F00 = g.reshape(3, -1).sum(axis=1) / self.N if self.bc_relax != 1.0 else np.zeros(3)
        tau = self.calc_stress(self.mu_0, g)
        F0 = np.zeros(3) if np.linalg.norm(self.BC_MQ) < EPS else tau.reshape(3, -1).sum(axis=1) / self.N
        T = self.g0_heat(self.mu_0, self.div_heat(tau), -1.0)
        R = -1.0 * (self.bc_relax * (self.BC_MQ @ F0) - (1 - self.bc_relax) * (self.BC_M @ (self.BC_QC0 @ F00)))
        return self.eps_heat(E, T) + R[:, None, None, None]

    def component_norm(self, g):
        return np.sqrt((g.reshape(3, -1) ** 2).sum(axis=1) / self.N)

    def run(self, E0, S0=None, P=None):
        """LSSolver::run -> runBasic  F:21247-21398, F:21716-21805 with the stop rule _converged  F:21177-21244
        and EpsilonErrorEstimator  F:14591-14637 (fix_dim zeroes entries 3..8 for dim 3, F:12122-12124)."""
        E0 = np.asarray(E0, dtype=np.float64)
        S0 = np.zeros(3) if S0 is None else np.asarray(S0, dtype=np.float64)
        if P is not None:
            self.set_bc_projector(P)
        self._bc_matrices()
        se = math.sqrt(EPS)
        if np.linalg.norm(self.BC_P @ S0) > se * np.linalg.norm(S0):
            raise RuntimeError("Incompatible stress boundary condition specified")
        if np.linalg.norm(self.BC_Q @ E0) > se * np.linalg.norm(E0):
            raise RuntimeError("Incompatible strain boundary condition specified")
        self.residuals = []
        self.error = None
        self.eps = np.zeros((3, self.nx, self.ny, self.nz))
        prev = float(np.linalg.norm(self.component_norm(self.eps)))
        it = 1
        update_ref = self.update_ref != "never"
        E = E0
        while True:
            if update_ref:
                self.calc_ref_material()
                self._bc_matrices()
                E = self.calc_bc_mean(E0, S0)
                update_ref = False
            self.eps = self.basic_scheme(E, self.eps)
            cur = float(np.linalg.norm(self.component_norm(self.eps)))
            abs_err = abs(prev - cur)
            rel_err = abs_err / (SMALLEST + cur)
            prev = cur
            if math.isnan(rel_err):
                self.error = "NaN detected in solution. Aborting."
                return True
            self.residuals.append(rel_err)
            if self.callback is not None and self.callback():
                break
            if it >= self.maxiter:
                break
            if rel_err <= self.tol or abs_err <= self.abs_tol:
                if self.bc_error(E0, S0) <= self.bc_tol:
                    break
            it += 1
        self.iterations = it
        return False

    def inner_l2(self, a, b, c=None):
        """innerProductL2, dim 3 branch  F:20955-20980 (three-argument form a:(b-c)  F:20871-20953): plain sum / N"""
        d = b if c is None else (b - c)
        return float((a[0] * d[0] + a[1] * d[1] + a[2] * d[2]).sum()) / self.N

    def run_cg(self, E0):
        """LSSolver::run with method=cg in the scalar modes: runCG -> runCGElasticity  F:22056-22066, F:23153-23247
        (the routine is dimension-generic: krylovOperator = one basic-scheme pass with E = 0)."""
        E0 = np.asarray(E0, dtype=np.float64)
        self.residuals = []
        self.error = None
        self.eps = np.zeros((3, self.nx, self.ny, self.nz))
        if self.update_ref != "never":
            self.calc_ref_material()
        prev = float(np.linalg.norm(self.component_norm(self.eps)))
        Z = np.zeros(3)
        eps = np.empty_like(self.eps)
        eps[:] = E0[:, None, None, None]
        r = self.basic_scheme(Z, eps)
        r = r + (E0[:, None, None, None] - eps)
        gamma = self.inner_l2(r, r) + SMALLEST
        gamma0 = gamma
        p = r.copy()
        it = 0
        while True:
            w = self.basic_scheme(Z, p)
            alpha = self.inner_l2(p, p, w) + SMALLEST
            alpha = gamma / alpha
            eps = eps + alpha * p
            self.eps = eps
            if getattr(self, "error_estimator", "epsilon") == "residual":
                # ResidualErrorEstimator::update_cg  F:14382-14405
                abs_err = math.sqrt(gamma)
                rel_err = math.sqrt(gamma / gamma0)
            else:
                cur = float(np.linalg.norm(self.component_norm(eps)))
                abs_err = abs(prev - cur)
                rel_err = abs_err / (SMALLEST + cur)
                prev = cur
            if math.isnan(rel_err):
                self.error = "NaN detected in solution. Aborting."
                return True
            self.residuals.append(rel_err)
            if self.callback is not None and self.callback():
                break
            if it >= self.maxiter:
                break
            if rel_err <= self.tol or abs_err <= self.abs_tol:
                if self.bc_error(E0) <= self.bc_tol:
                    break
            it += 1
            r = r + (-alpha) * (p - w)
            delta = self.inner_l2(r, r) + SMALLEST
            beta = delta / gamma
            gamma = delta
            p = r + beta * p
        self.iterations = it
        return False

    def bc_error(self, E_cur, S_cur=None):
        """bc_error  F:21129-21161 (dim 3: plain norms)"""
        S_cur = np.zeros(3) if S_cur is None else np.asarray(S_cur, dtype=np.float64)
        Emean, Smean = self.mean_strain(), self.mean_stress()
        nE = float(np.linalg.norm(self.BC_P @ E_cur))
        err_F = float(np.linalg.norm(self.BC_P @ Emean - E_cur)) / (1 if nE < self.bc_tol else nE)
        nS = float(np.linalg.norm(S_cur))
        err_S = float(np.linalg.norm(self.BC_Q @ Smean - S_cur)) / (1 if nS < self.bc_tol else nS)
        return max(err_F, err_S)

    def calc_effective_properties(self):
        """calc_effective_properties, heat / porous branch  F:26115-26165: three unit gradients,
        Ceff = S E^-1 (3x3, no Voigt halving)."""
        S = np.zeros((3, 3))
        self.ceff_iterations = []
        for i in range(3):
            Ep = np.zeros(3)
            Ep[i] = 1.0
            if self.run(Ep):
                raise RuntimeError(self.error)
            S[:, i] = self.mean_stress()
            self.ceff_iterations.append(self.iterations)
        return S
