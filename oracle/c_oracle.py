"""ctypes wrapper of oracle/c/libfg_ref.so: the reference's per-iteration loop nests in
C/OpenMP with the FFT supplied by scipy.fft (pocketfft, `workers` threads) in place of
threaded FFTW.  TEST INFRASTRUCTURE ONLY (checker + bench.py cpu_baseline)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import scipy.fft

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)


def _P(a):
    # the C side indexes [c][x][y][z] with z fastest: a Fortran-ordered or strided array (np.stack of transposed views
    # keeps the inputs' order) must not reach it silently
    if not (a.flags["C_CONTIGUOUS"] and a.dtype == np.float64):
        raise ValueError("oracle/c expects C-contiguous float64 arrays")
    return a.ctypes.data_as(_dp)


def load(build=True, native=False):
    """native: the reference's release flags (-O3 -march=native), compiled on this very host (cpu_baseline only)."""
    name = "libfg_ref_native.so" if native else "libfg_ref.so"
    so = os.path.join(_HERE, "c", name)
    fresh = native and os.environ.get("FG_REF_NATIVE_BUILT") == "1" and os.path.exists(so)   # built by the parent process on this host
    if (native and not fresh) or (not os.path.exists(so) and build):
        # the native library is always rebuilt: one made on another machine may carry instructions this host lacks
        if native and os.path.exists(so):
            os.remove(so)
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), name], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.ref_max_threads.restype = ctypes.c_int
    return lib


class CRef:
    """One pass of basicScheme (F:20558-20578) with the reference's pass structure."""

    def __init__(self, n, dims, mats, phis, normals=None, mixing="voigt", threads=None):
        self.lib = load()
        self.nx, self.ny, self.nz = n
        self.dims = tuple(float(d) for d in dims)
        self.N = self.nx * self.ny * self.nz
        self.mu = np.array([m[0] for m in mats], dtype=np.float64)
        self.lam = np.array([m[1] for m in mats], dtype=np.float64)
        self.phi = np.ascontiguousarray(np.stack(phis), dtype=np.float64)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
        self.mixing = {"voigt": 0, "laminate": 1}[mixing]
        self.threads = threads or self.lib.ref_max_threads()
        self.lib.ref_set_threads(int(self.threads))
        self.eps_g = np.finfo(float).eps
        self.eps_a = np.finfo(float).eps ** (2.0 / 3.0)
        self.fft_seconds = 0.0

    def _d(self, v):
        return ctypes.c_double(float(v))

    def calc_stress(self, mu_0, lambda_0, eps, alpha=1.0):
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        tau = np.empty_like(eps)
        err = self.lib.ref_calc_stress(self.nx, self.ny, self.nz, _P(eps), _P(self.phi),
                                       _P(self.normals) if self.normals is not None else None, len(self.mu),
                                       _P(self.mu), _P(self.lam), self.mixing, self._d(mu_0), self._d(lambda_0),
                                       self._d(alpha), self._d(self.eps_g), self._d(self.eps_a), _P(tau))
        if err:
            raise RuntimeError("The laminate mixing rule supports only two phase mixtures")
        return tau

    def mean_stress(self, eps):
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        out = np.zeros(6)
        self.lib.ref_mean_stress(self.nx, self.ny, self.nz, _P(eps), _P(self.phi),
                                 _P(self.normals) if self.normals is not None else None, len(self.mu), _P(self.mu),
                                 _P(self.lam), self.mixing, self._d(self.eps_g), self._d(self.eps_a), _P(out))
        return out

    def div(self, tau):
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        f = np.empty((3,) + tau.shape[1:])
        self.lib.ref_div(self.nx, self.ny, self.nz, *map(self._d, self.dims), _P(tau), _P(f))
        return f

    def g0(self, mu_0, lambda_0, fh, alpha):
        fh = np.ascontiguousarray(fh, dtype=np.complex128)
        self.lib.ref_g0(self.nx, self.ny, self.nz, *map(self._d, self.dims), self._d(mu_0), self._d(lambda_0),
                        self._d(alpha), fh.ctypes.data_as(ctypes.c_void_p))
        return fh

    def eps_op(self, E, u):
        E = np.ascontiguousarray(E, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        y = np.empty((6,) + u.shape[1:])
        self.lib.ref_eps(self.nx, self.ny, self.nz, *map(self._d, self.dims), _P(E), _P(u), _P(y))
        return y

    def component_norm(self, eps):
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        m = np.zeros(6)
        self.lib.ref_component_norm(ctypes.c_size_t(self.N), _P(eps), _P(m))
        return m

    def basic_scheme(self, E, eps, mu_0, lambda_0):
        import time
        tau = self.calc_stress(mu_0, lambda_0, eps)
        f = self.div(tau)
        t = time.perf_counter()
        fh = scipy.fft.rfftn(f, axes=(1, 2, 3), workers=self.threads)     # 3 x FFTW r2c  F:18496-18499
        self.fft_seconds += time.perf_counter() - t
        v = fh.view(np.float64)
        self.lib.ref_scale(ctypes.c_size_t(v.size), self._d(1 / float(self.N)), _P(v))  # F:18501-18506
        fh = self.g0(mu_0, lambda_0, fh, -1.0)
        t = time.perf_counter()
        u = scipy.fft.irfftn(fh, s=(self.nx, self.ny, self.nz), axes=(1, 2, 3), workers=self.threads, norm="forward")
        self.fft_seconds += time.perf_counter() - t
        u = np.ascontiguousarray(u)
        e = self.eps_op(E, u)
        self.lib.ref_add(ctypes.c_size_t(self.N), _P(np.zeros(6)), _P(e))  # applyBCProjector's eps.add(R)  F:20269
        return e


class CRefCG(CRef):
    """runCGElasticity  F:23153-23247 on the C loop nests (prescribed mean strain: calcBCMean returns E0, bc_error = 0):
    krylovOperator F:20583-20587 = basicScheme with E = 0, innerProductL2 F:20871-21038, xpay F:9819-9838, xpaymz
    F:9993-10010, adjustResidual F:10012-10022; error estimators: epsilon F:14591-14637, residual F:14382-14405.
    The checker of the GPU's CG at BASELINE's sizes (tests/test_gpu_fullsize_oracle.py); itself held against
    LSOracle._run_cg_step in tests/test_c_oracle.py."""

    def inner_l2(self, a, b, c=None):
        self.lib.ref_inner_l2.restype = ctypes.c_double
        return float(self.lib.ref_inner_l2(self.nx, self.ny, self.nz, 6, _P(a), _P(b), _P(c) if c is not None else None))

    def run_cg(self, E, mu_0, lambda_0, maxiter, tol=0.0, abs_tol=0.0, estimator="epsilon", start_norm=0.0):
        """Returns (eps, residual history, iteration count as LSSolver::run reports it)."""
        import math
        lib, d, sz = self.lib, ctypes.c_double, ctypes.c_size_t
        small = float(np.finfo(float).tiny)
        E = np.ascontiguousarray(E, dtype=np.float64)
        Z = np.zeros(6)
        shape = (6, self.nx, self.ny, self.nz)
        n6 = 6 * self.N
        eps = np.empty(shape)
        lib.ref_set_constant(sz(self.N), 6, _P(E), _P(eps))            # epsilon.setConstant(E)  F:23183
        r = self.basic_scheme(Z, eps, mu_0, lambda_0)                  # krylovOperator
        lib.ref_adjust_residual(sz(self.N), 6, _P(E), _P(eps), _P(r))
        gamma = self.inner_l2(r, r) + small
        gamma0 = gamma
        p = r.copy()
        prev = start_norm      # EpsilonErrorEstimator's norm of the field the step started from (zero field: 0)
        residuals, it = [], 0
        while True:
            w = self.basic_scheme(Z, p, mu_0, lambda_0)
            alpha = self.inner_l2(p, p, w) + small
            alpha = gamma / alpha
            lib.ref_xpay(sz(n6), _P(eps), d(alpha), _P(p), _P(eps))
            if estimator == "residual":
                abs_err, rel_err = math.sqrt(gamma), math.sqrt(gamma / gamma0)
            else:
                m = self.component_norm(eps)
                cur = math.sqrt(float((m * m).sum() + (m[3:] * m[3:]).sum()))
                abs_err = abs(prev - cur)
                rel_err = abs_err / (small + cur)
                prev = cur
            residuals.append(rel_err)
            if it >= maxiter or rel_err <= tol or abs_err <= abs_tol:
                break
            it += 1
            lib.ref_xpaymz(sz(n6), _P(r), d(-alpha), _P(p), _P(w), _P(r))
            delta = self.inner_l2(r, r) + small
            beta = delta / gamma
            gamma = delta
            lib.ref_xpay(sz(n6), _P(r), d(beta), _P(p), _P(p))
        return eps, residuals, it


class CRefScalar:
    """One pass of basicScheme in the scalar modes (heat / porous, BASELINE config 5): GammaOperatorStaggeredHeat
    F:20342-20351 with prescribed mean gradients -- polarisation, divOperatorStaggeredHeat, r2c + 1/N, the heat Green
    operator, c2r, epsOperatorStaggeredHeat -- on the C loop nests (checker at 256^3, where the NumPy oracle is slow)."""

    def __init__(self, n, dims, mus, phis, threads=None):
        self.lib = load()
        self.nx, self.ny, self.nz = n
        self.dims = tuple(float(d) for d in dims)
        self.N = self.nx * self.ny * self.nz
        self.mu = np.ascontiguousarray(mus, dtype=np.float64)
        self.phi = np.ascontiguousarray(np.stack(phis), dtype=np.float64)
        self.threads = threads or self.lib.ref_max_threads()
        self.lib.ref_set_threads(int(self.threads))

    def basic_scheme(self, E, g, mu_0):
        d = ctypes.c_double
        lib = self.lib
        n3 = (self.nx, self.ny, self.nz)
        E = np.ascontiguousarray(E, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        tau = np.empty_like(g)
        lib.ref_calc_stress_scalar(*n3, _P(g), _P(self.phi), len(self.mu), _P(self.mu), d(mu_0), d(1.0), _P(tau))
        f = np.empty(n3)
        lib.ref_div_heat(*n3, *map(d, self.dims), _P(tau), _P(f))
        th = scipy.fft.rfftn(f, workers=self.threads)                      # fftVector(., 1)  F:18481-18510
        v = th.view(np.float64)
        lib.ref_scale(ctypes.c_size_t(v.size), d(1 / float(self.N)), _P(v))
        lib.ref_g0_heat(*n3, *map(d, self.dims), d(mu_0), d(-1.0), th.ctypes.data_as(ctypes.c_void_p))
        T = np.ascontiguousarray(scipy.fft.irfftn(th, s=n3, workers=self.threads, norm="forward"))
        out = np.empty_like(g)
        lib.ref_eps_heat(*n3, *map(d, self.dims), _P(E), _P(T), _P(out))
        return out


class CRefViscosity(CRef):
    """One pass of basicScheme in mode = viscosity on the C loop nests: DeltaOperatorStaggered F:20422-20460 composed of
    the elasticity routines (the law ScalarLinearIsotropicMaterialLaw(6) with mu / 2, F:15234-15239, is S = E * (alpha mu / 2):
    Hooke with (mu / 4, 0) performs the same multiplications, scalings by powers of two being exact)."""

    def __init__(self, n, dims, mus, phis, threads=None):
        super().__init__(n, dims, [(0.25 * m, 0.0) for m in mus], phis, None, "voigt", threads)

    def basic_scheme(self, E, eps, mu_0, lambda_0=0.0):
        alpha = -1.0
        m = 1 / (4 * mu_0)
        tau = self.calc_stress(mu_0, lambda_0, eps)
        adj = np.asarray(E, dtype=np.float64) - 2 * alpha * m * (tau.reshape(6, -1).sum(axis=1) / self.N)
        f = self.div(tau)
        fh = scipy.fft.rfftn(f, axes=(1, 2, 3), workers=self.threads)
        v = fh.view(np.float64)
        self.lib.ref_scale(ctypes.c_size_t(v.size), self._d(1 / float(self.N)), _P(v))
        fh = self.g0(-1.0 / (4 * m), float("inf"), fh, alpha)               # lambda_0 = inf: c20 = c10  F:19749-19755
        u = np.ascontiguousarray(scipy.fft.irfftn(fh, s=(self.nx, self.ny, self.nz), axes=(1, 2, 3), workers=self.threads,
                                                  norm="forward"))
        eta = self.eps_op(adj, u)
        eta += (2 * alpha * m) * tau                                         # eta.xpay(eta, 2 alpha m, tau_copy)
        return eta


class CRefLoop:
    """The loop of basicScheme as the reference runs it: ONE strain field that every routine works on in place
    (tau aliases epsilon, F:15153-15155, F:20558-20578), buffers allocated once, `threads` OpenMP threads
    (the reference's <num_threads>, default 1, F:25226).  cpu_baseline only; the checker is CRef.

    fft = "own" (default where the grid is a power of two): the threaded row-column transform of oracle/c/fg_fft_ref.c, so that
    every thread of a pass is an OpenMP thread (OMP_PROC_BIND / OMP_PLACES pin all of them) and the work fields are first
    touched by the threads that sweep them; "pocketfft": scipy.fft with `workers` threads (rounds 2-4)."""

    def __init__(self, n, dims, mats, phis, normals=None, mixing="voigt", threads=1, native=True, fft="own", loops="reference"):
        """loops = "reference": the stencil operators in the reference's traversal orders (z outermost in the x- and y-difference
        nests, F:18864-18887, F:18646-18675); "contiguous": the same values with z innermost everywhere (ref_div_contig /
        ref_eps_contig: what the traversal order costs, reported beside the baseline, never as it)."""
        self.lib = load(native=native)
        self.contig = loops == "contiguous"
        lib = self.lib
        self.nx, self.ny, self.nz = n
        self.dims = tuple(float(d) for d in dims)
        self.N = self.nx * self.ny * self.nz
        self.mu = np.array([m[0] for m in mats], dtype=np.float64)
        self.lam = np.array([m[1] for m in mats], dtype=np.float64)
        self.mixing = {"voigt": 0, "laminate": 1}[mixing]
        self.threads = int(threads)
        lib.ref_set_threads(self.threads)
        lib.ref_fft_plan.restype = ctypes.c_void_p
        self.own_fft = fft == "own" and bool(lib.ref_fft_supported(*n))
        self.plan = ctypes.c_void_p(lib.ref_fft_plan(*n)) if self.own_fft else None
        plane = self.ny * self.nz

        def placed(ncomp, src=None, plane_elems=plane):
            # pages placed by the threads that sweep them (ref_first_touch: per component a static loop over the x-planes)
            a = np.empty((ncomp, self.nx, plane_elems))
            lib.ref_first_touch(_P(a), ncomp, self.nx, ctypes.c_size_t(plane_elems))
            if src is not None:
                a[:] = np.asarray(src, dtype=np.float64).reshape(a.shape)
            return a
        self.phi = placed(len(phis), np.stack(phis)).reshape((len(phis),) + tuple(n))
        self.normals = None if normals is None else placed(3, normals).reshape((3,) + tuple(n))
        self.eps = placed(6).reshape((6,) + tuple(n))       # the one strain / polarisation field
        self.f = placed(3).reshape((3,) + tuple(n))         # divergence / displacement work field
        nzc = self.nz // 2 + 1
        self.fh = placed(3, None, 2 * self.ny * nzc).view(np.complex128).reshape(3, self.nx, self.ny, nzc) if self.own_fft else None
        self.zero6 = np.zeros(6)
        self.norms = np.zeros(6)
        self.eps_g = np.finfo(float).eps
        self.eps_a = np.finfo(float).eps ** (2.0 / 3.0)
        self.fft_seconds = 0.0

    def one_pass(self, E, mu_0, lambda_0):
        import time
        d = ctypes.c_double
        lib = self.lib
        lib.ref_set_threads(self.threads)
        E = np.ascontiguousarray(E, dtype=np.float64)
        nrm = _P(self.normals) if self.normals is not None else None
        # calcStressDiff in place: tau over epsilon  F:18030-18033
        if lib.ref_calc_stress(self.nx, self.ny, self.nz, _P(self.eps), _P(self.phi), nrm, len(self.mu), _P(self.mu),
                               _P(self.lam), self.mixing, d(mu_0), d(lambda_0), d(1.0), d(self.eps_g), d(self.eps_a),
                               _P(self.eps)):
            raise RuntimeError("The laminate mixing rule supports only two phase mixtures")
        (lib.ref_div_contig if self.contig else lib.ref_div)(self.nx, self.ny, self.nz, *map(d, self.dims), _P(self.eps), _P(self.f))
        t = time.perf_counter()
        if self.own_fft:
            fh = self.fh
            lib.ref_fft_r2c(self.plan, 3, _P(self.f), fh.ctypes.data_as(_dp))
        else:
            fh = scipy.fft.rfftn(self.f, axes=(1, 2, 3), workers=self.threads)
        self.fft_seconds += time.perf_counter() - t
        v = fh.view(np.float64)
        lib.ref_scale(ctypes.c_size_t(v.size), d(1 / float(self.N)), v.ctypes.data_as(_dp))
        lib.ref_g0(self.nx, self.ny, self.nz, *map(d, self.dims), d(mu_0), d(lambda_0), d(-1.0), fh.ctypes.data_as(ctypes.c_void_p))
        t = time.perf_counter()
        if self.own_fft:
            lib.ref_fft_c2r(self.plan, 3, fh.ctypes.data_as(_dp), _P(self.f))
            u = self.f
        else:
            u = np.ascontiguousarray(scipy.fft.irfftn(fh, s=(self.nx, self.ny, self.nz), axes=(1, 2, 3), workers=self.threads,
                                                      norm="forward", overwrite_x=True))
        self.fft_seconds += time.perf_counter() - t
        (lib.ref_eps_contig if self.contig else lib.ref_eps)(self.nx, self.ny, self.nz, *map(d, self.dims), _P(E), _P(u), _P(self.eps))
        lib.ref_add(ctypes.c_size_t(self.N), _P(self.zero6), _P(self.eps))          # applyBCProjector's eps.add(R)  F:20269
        lib.ref_component_norm(ctypes.c_size_t(self.N), _P(self.eps), _P(self.norms))
        return self.norms
