"""NumPy restatement of fibergen's basic scheme in mode=viscosity (dual Stokes scheme).

TEST INFRASTRUCTURE ONLY -- the checker, never the product path (see oracle/ls_oracle.py).

The fluid problem is solved in dual form: the 6-component field "epsilon" holds the (traceless) fluid stress,
the constitutive law gives the shear rate  S = (fluidity / 2) E  (ScalarLinearIsotropicMaterialLaw(6) with
mu *= 0.5, F:15234-15239), and Gamma0 is replaced by the operator of DeltaOperatorStaggered F:20422-20460
(the staggered Green operator with lambda0 -> infinity, i.e. the projection onto divergence-free velocity
fields, plus a multiple of the identity).  ``F`` = /root/reference/src/fibergen.cpp.

Pinning (tests/test_oracle_pins.py::test_viscosity_*): no numeric fixtures exist in the reference for this
mode; pinned on closed forms the discretisation reproduces exactly: the homogeneous fluid, layered fluids
(shear stress across the layers is uniform -> arithmetic mean of the fluidities; in-plane shear rate is
uniform -> harmonic mean), and the preservation of zero trace (incompressibility) by the operator.
"""
from __future__ import annotations

import math

import numpy as np

from .ls_oracle import EPS, SMALLEST, LSOracle

VOIGT_THRESHOLD = 10 * EPS


class ViscosityOracle(LSOracle):
    """mats = [(mu, 0.0)] with mu the XML constant (before the halving of F:15237)."""

    def pk1(self, eps, alpha=1.0):
        """VoigtMixedMaterialLaw<.,.,6>::PK1  F:12752-12761 over ScalarLinearIsotropicMaterialLaw::PK1
        F:11182-11198 with law->mu = 0.5 mu (F:15237)."""
        if self.mixing_rule != "voigt":
            raise RuntimeError("viscosity restatement: Voigt mixing only")
        P = np.zeros_like(eps)
        first = np.ones(eps.shape[1:], dtype=bool)
        for phi, (mu, _lam) in zip(self.phis, self.mats):
            live = phi > VOIGT_THRESHOLD
            alpha_mu = (phi * alpha) * (mu * 0.5)
            for m in range(6):
                term = eps[m] * alpha_mu
                P[m] = np.where(live, np.where(first, term, P[m] + term), P[m])
            first &= ~live
        return P

    def _tangent_moduli(self):
        raise NotImplementedError

    def tangent_eig_minmax(self):
        """getRefMaterial(zero_trace=true)  F:12153-12236 / eig  F:12496-12509: the tangent is
        (sum_p phi_p mu_p / 2) Id6, every eigenvalue of its 5x5 sub-block equals that factor."""
        t = np.zeros((self.nx, self.ny, self.nz))
        first = np.ones(t.shape, dtype=bool)
        for phi, (mu, _lam) in zip(self.phis, self.mats):
            live = phi > VOIGT_THRESHOLD
            v = (phi * 1.0) * (mu * 0.5)
            t = np.where(live, np.where(first, v, t + v), t)
            first &= ~live
        return float(t.min()), float(t.max())

    def delta_staggered(self, E, mu_0, lambda_0, tau, alpha=-1.0):
        """DeltaOperatorStaggered  F:20422-20460"""
        m = 1 / (4 * mu_0)                                  # fluidity -> viscosity
        tau_copy = tau.copy()
        adj = E - 2 * alpha * m * (tau_copy.reshape(6, -1).sum(axis=1) / self.N)
        eta = self.gamma_staggered(adj, -1.0 / (4 * m), math.inf, tau, alpha)
        return eta + (2 * alpha * m) * tau_copy               # eta.xpay(eta, 2 alpha mu_0, tau_copy)

    def basic_scheme(self, E, eps):
        """basicScheme  F:20558-20578 -> GammaOperator, viscosity branch  F:20515-20518"""
        self._F00 = np.zeros(6)
        tau = self.calc_stress(self.mu_0, self.lambda_0, eps)
        return self.delta_staggered(np.asarray(E, dtype=np.float64), self.mu_0, self.lambda_0, tau, -1.0)

    def velocity(self):
        """get_raw_field("u"), viscosity branch  F:15528-15535"""
        tau = self.calc_stress(self.mu_0, self.lambda_0, self.eps)
        return self.g0_staggered(1 / (4 * self.mu_0), math.inf, self.div_staggered(tau), 1 / (2 * self.mu_0))

    # run_cg is inherited: runCG sends every non-hyperelastic mode to runCGElasticity (F:22056-22066), whose Krylov
    # operator is one basic-scheme pass with E = 0 -- here the Delta operator above.
