"""Oracle parity AT BASELINE.json's sizes on the benchmark workload itself (fibergen_amd.rve.bench_rve: the RVE
bench.py times), through the C ABI.  The checker is oracle/c (CRef: the reference's loop nests in C/OpenMP + pocketfft,
bit-identical per stage to the NumPy oracle, tests/test_c_oracle.py) because the NumPy oracle needs minutes per pass
at these sizes.

  256^3  Voigt and laminate: three passes from a non-trivial strain field -- pass 1 runs the strain-state pipeline,
         passes 2-3 the default displacement loop (tiled sweep, fused x pass): eps <= 1e-11, sums of squares <= 1e-12
  512^3  laminate (BASELINE config 3): two passes, the same bars (skipped when the host has too little memory)
  128^3  (BASELINE config 2) converged runs at tol 1e-6, Voigt and laminate: iteration count equal, residual history,
         the effective-stiffness column <S> <= 1e-9, strain field <= 1e-9
"""
import math

import numpy as np
import pytest

from helpers import INCLUSION, MATRIX, lame, rel_err

pytestmark = pytest.mark.gpu

E_LOAD = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.5])
DIMS = (1.0, 1.0, 1.0)


def _workload(n, mixing):
    from fibergen_amd.rve import bench_rve
    phi, normals, _ = bench_rve(n, mixing)
    return [lame(**MATRIX), lame(**INCLUSION)], [1.0 - phi, phi], normals


def _gpu(n, mats, phis, normals, mixing, **kw):
    from fibergen_amd import LSSolver
    s = LSSolver(n, n, n)
    s.set_num_phases(2)
    for p in range(2):
        s.set_phase(p, mats[p][0], mats[p][1], phis[p])
    if normals is not None:
        s.set_normals(normals)
    s.set_options(mixing_rule=mixing, **kw)
    return s


def _cref(n, mats, phis, normals, mixing):
    import os
    from oracle.c_oracle import CRef
    # 16 threads: the best point of the host's thread sweep (bench.py cpu_baseline); all 256 are 5x slower
    return CRef((n, n, n), DIMS, mats, phis, normals, mixing, threads=min(16, os.cpu_count() or 1))


def _start_field(n, phi):
    """a deterministic, non-trivial strain field: the load plus phase- and position-dependent perturbations"""
    x = (np.arange(n) + 0.5) / n
    w = np.sin(2 * np.pi * x)[:, None, None] * np.cos(4 * np.pi * x)[None, :, None] + 0.5 * np.sin(6 * np.pi * x)[None, None, :]
    eps = np.empty((6, n, n, n))
    for c in range(6):
        eps[c] = E_LOAD[c] + 0.05 * (c + 1) * phi + 0.02 * (6 - c) * w
    return eps


def _passes_against_cref(n, mixing, passes, **opts):
    mats, phis, normals = _workload(n, mixing)
    s = _gpu(n, mats, phis, normals, mixing, **opts)
    mu_0, lam_0 = s.calc_ref_material()
    # closed form of calcRefMaterial for isotropic phases with phi covering [0, 1]  (F:22283-22313, F:12763-12771):
    # tangent eigenvalues {2 mu, 2 mu + 3 lambda} of the mixture, extremes at the pure phases
    eig = [v for mu, lam in mats for v in (2 * mu, 2 * mu + 3 * lam)]
    assert mu_0 == pytest.approx(0.25 * (min(eig) + max(eig)), rel=1e-14) and lam_0 == 0.0
    eps0 = _start_field(n, phis[1])
    s.set_field("epsilon", eps0)
    s.iterate(E_LOAD, passes)            # pass 1: strain-state pipeline; then the displacement loop
    got = s.get_field("epsilon")
    sumsq = s.get_field("sumsq")          # norm sweep of the last displacement pass: belongs to eps_{passes-1}
    s.close()
    c = _cref(n, mats, phis, normals, mixing)
    eps = eps0
    prev = None
    for _ in range(passes):
        prev = eps
        eps = c.basic_scheme(E_LOAD, eps, mu_0, lam_0)
    assert rel_err(got, eps) < 1e-11
    want_sumsq = (c.component_norm(prev) ** 2) * float(n) ** 3
    assert np.abs(sumsq / want_sumsq - 1).max() < 1e-12


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_bench_workload_256_three_passes(mixing):
    _passes_against_cref(256, mixing, 3)


@pytest.mark.parametrize("n,mixing", [(200, "voigt"), (200, "laminate"), (300, "voigt")])
def test_bench_workload_decimal_grids_three_passes(n, mixing):
    """The decimal grid sizes (200 = 20 x 10, 300 = 20 x 15; nz / 2 = 100, 150): every transform pass runs the Stockham tile
    kernels of fg_fft_smooth.h, the x pass unfused; the checker transforms with pocketfft.  Same bars as at 256^3."""
    _passes_against_cref(n, mixing, 3)


def test_bench_workload_512_laminate_two_passes():
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 96 * 2 ** 30:
        pytest.skip("needs ~64 GB of free host memory for the 512^3 checker")
    _passes_against_cref(512, "laminate", 2)


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_config2_converged_run_128(mixing):
    """BASELINE config 2: 128^3 fibre RVE, tol 1e-6 -- LSSolver::run on the GPU against the same loop on the checker
    (runBasic F:21716-21805 with the stop rule of _converged F:21177-21244; pure strain BC, so bc_error = 0)."""
    n = 128
    mats, phis, normals = _workload(n, mixing)
    s = _gpu(n, mats, phis, normals, mixing, tol=1e-6)
    assert s.run(E_LOAD) is False
    mu_0, lam_0 = s.ref_material
    c = _cref(n, mats, phis, normals, mixing)
    eps = np.zeros((6, n, n, n))
    prev, it, residuals = 0.0, 1, []
    tiny = np.finfo(float).tiny
    while True:
        eps = c.basic_scheme(E_LOAD, eps, mu_0, lam_0)
        m = c.component_norm(eps)
        cur = math.sqrt(float((m * m).sum() + (m[3:] * m[3:]).sum()))   # 9 mirrored entries  F:14600-14609
        abs_err = abs(prev - cur)
        rel = abs_err / (tiny + cur)
        prev = cur
        residuals.append(rel)
        if rel <= 1e-6 or abs_err <= np.finfo(float).eps or it >= 10000:
            break
        it += 1
    assert s.iterations == it
    assert np.abs(np.array(s.residuals) - np.array(residuals)).max() < 1e-11
    assert rel_err(s.get_field("epsilon"), eps) < 1e-9
    assert rel_err(s.mean_stress(), c.mean_stress(eps)) < 1e-9       # the Ceff column of this load case
    s.close()


# ---- BASELINE config 5 at its size: 256^3 porous (scalar mode) and viscosity (dual Stokes scheme) on the benchmark RVE,
#      three passes from a non-trivial field against the C loop nests (oracle/c: CRefScalar, CRefViscosity)
def _threads():
    import os
    return min(16, os.cpu_count() or 1)


def test_config5_porous_256_four_passes():
    from fibergen_amd import LSSolver
    from fibergen_amd.rve import bench_rve
    from oracle.c_oracle import CRefScalar
    n = 256
    phi, _, _ = bench_rve(n, "voigt")
    mus, phis = [1.0, 10.0], [1.0 - phi, phi]      # bench.py's porous materials
    E = np.array([1.0, 0.0, 0.0])
    s = LSSolver(n, n, n)
    s.set_options(mode="porous")
    s.set_num_phases(2)
    for p in range(2):
        s.set_phase(p, mus[p], 0.0, phis[p])
    # the state of the scalar modes is the potential (fields cannot be set): four passes of LSSolver::run from the zero
    # field, i.e. g_1 = E, then three passes with the full operator
    s.set_options(tol=0.0, abs_tol=0.0, maxiter=4)
    assert s.run(E) is False and s.iterations == 4
    mu_0, _ = s.ref_material
    got = s.get_field("epsilon")
    res = np.array(s.residuals)
    s.close()
    c = CRefScalar((n, n, n), DIMS, mus, phis, threads=_threads())
    g = np.zeros((3, n, n, n))
    prev, want = 0.0, []
    for _ in range(4):
        g = c.basic_scheme(E, g, mu_0)
        cur = math.sqrt(float((g.reshape(3, -1) ** 2).sum(axis=1).sum() / n ** 3))   # EpsilonErrorEstimator F:14591-14637, dim 3
        want.append(abs(prev - cur) / (np.finfo(float).tiny + cur))
        prev = cur
    assert rel_err(got, g) < 1e-11
    assert np.abs(res - np.array(want)).max() < 1e-12


def test_config5_viscosity_256_three_passes():
    from fibergen_amd import LSSolver
    from fibergen_amd.rve import bench_rve
    from oracle.c_oracle import CRefViscosity
    n = 256
    phi, _, _ = bench_rve(n, "voigt")
    mus, phis = [1.0, 0.1], [1.0 - phi, phi]       # bench.py's fluid with ten times more viscous particles
    E = np.array([1.0, -1.0, 0.0, 0.0, 0.0, 0.0])   # traceless prescribed stress (F:26257-26261)
    s = LSSolver(n, n, n)
    s.set_options(mode="viscosity")
    s.set_num_phases(2)
    for p in range(2):
        s.set_phase(p, mus[p], 0.0, phis[p])
    mu_0, lam_0 = s.calc_ref_material()
    e0 = _start_field(n, phi)
    e0[:3] -= e0[:3].mean(axis=0)                  # the field of this mode is a traceless stress
    s.set_field("epsilon", e0)
    s.iterate(E, 3)
    got = s.get_field("epsilon")
    s.close()
    c = CRefViscosity((n, n, n), DIMS, mus, phis, threads=_threads())
    eps = e0
    for _ in range(3):
        eps = c.basic_scheme(E, eps, mu_0, lam_0)
    assert rel_err(got, eps) < 1e-11


# ---- the conjugate gradients at the sizes bench.py times them (runCGElasticity F:23153-23247): `iters` CG iterations from the
#      zero field through LSSolver::run (fg_run_load_case) against the same loop on the C loop nests (oracle/c_oracle.CRefCG,
#      held against the NumPy oracle per iteration in tests/test_c_oracle.py); fused tiled sweeps (k_cgu_tile, k_u_tile<..CGP>)
#      and the four-kernel form
def _cg_against_cref(n, mixing, iters, fused_forms=(1, 0), estimator="epsilon"):
    from oracle.c_oracle import CRefCG
    import os
    mats, phis, normals = _workload(n, mixing)
    got = {}
    for fused in fused_forms:
        s = _gpu(n, mats, phis, normals, mixing, method="cg", tol=0.0, abs_tol=0.0, maxiter=iters, cg_fused=fused,
                 error_estimator=estimator)
        assert s.run(E_LOAD) is False
        assert s.iterations == iters
        got[fused] = (s.ref_material, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress().copy())
        s.close()
    (mu_0, lam_0) = got[fused_forms[0]][0]
    c = CRefCG((n, n, n), DIMS, mats, phis, normals, mixing, threads=min(16, os.cpu_count() or 1))
    eps, residuals, it = c.run_cg(E_LOAD, mu_0, lam_0, maxiter=iters, estimator=estimator)
    assert it == iters
    want_stress = c.mean_stress(eps)
    for fused in fused_forms:
        _, res, e, ms = got[fused]
        assert res.shape == (iters + 1,)
        assert np.abs(res - np.array(residuals)).max() < 1e-11
        assert rel_err(e, eps) < 1e-10
        assert rel_err(ms, want_stress) < 1e-10


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_cg_256_four_iterations_fused_and_unfused(mixing):
    _cg_against_cref(256, mixing, 4)


def test_cg_256_residual_estimator():
    _cg_against_cref(256, "voigt", 3, fused_forms=(1,), estimator="residual")


def test_cg_512_laminate_three_iterations():
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 128 * 2 ** 30:
        pytest.skip("needs ~100 GB of free host memory for the 512^3 CG checker")
    _cg_against_cref(512, "laminate", 3, fused_forms=(1,))


# ---- large NON-cubic grids with mixed-radix axes (p 2^k lengths: 3 * 64, 5 * 32, 25 * 8) and every tile shape of the sweep
#      (nz/2 = 64: one wave per row, 128: two, 96: halo lanes), against the C loop nests like the cubes above
@pytest.mark.parametrize("grid,dims,mixing", [
    ((192, 160, 128), (1.0, 1.0, 1.0), "voigt"),       # x = 3 * 64, y = 5 * 32; a z row is one wave
    ((96, 200, 256), (1.5, 1.0, 2.0), "laminate"),     # y = 25 * 8; a z row is two waves; anisotropic cell
    ((160, 96, 192), (1.0, 2.0, 1.0), "laminate"),     # z = 2 * 96 = 2 * 3 * 32: packed rows of a mixed-radix length, halo-lane tiles
])
def test_large_noncubic_mixed_radix_grids_three_passes(grid, dims, mixing):
    from helpers import two_phase_setup
    from fibergen_amd import LSSolver
    from oracle.c_oracle import CRef
    import os
    mats, phis, normals = two_phase_setup(grid, mixing)
    s = LSSolver(*grid, *dims)
    s.set_num_phases(2)
    for p in range(2):
        s.set_phase(p, mats[p][0], mats[p][1], phis[p])
    s.set_normals(normals)
    s.set_options(mixing_rule=mixing)
    mu_0, lam_0 = s.calc_ref_material()
    x = [(np.arange(n) + 0.5) / n for n in grid]
    w = np.sin(2 * np.pi * x[0])[:, None, None] * np.cos(4 * np.pi * x[1])[None, :, None] + 0.5 * np.sin(6 * np.pi * x[2])[None, None, :]
    eps0 = np.empty((6,) + grid)
    for c in range(6):
        eps0[c] = E_LOAD[c] + 0.05 * (c + 1) * phis[1] + 0.02 * (6 - c) * w
    s.set_field("epsilon", eps0)
    s.iterate(E_LOAD, 3)
    got = s.get_field("epsilon")
    sumsq = s.get_field("sumsq")
    s.close()
    c = CRef(grid, dims, mats, phis, normals, mixing, threads=min(16, os.cpu_count() or 1))
    eps, prev = eps0, None
    for _ in range(3):
        prev = eps
        eps = c.basic_scheme(E_LOAD, eps, mu_0, lam_0)
    assert rel_err(got, eps) < 1e-11
    want = (c.component_norm(prev) ** 2) * float(np.prod(grid))
    assert np.abs(sumsq / want - 1).max() < 1e-12


def test_cg_on_large_noncubic_grid_three_iterations():
    """runCGElasticity on a mixed-radix non-cubic grid (y = 25 * 8, two waves per z row, laminate mixing, anisotropic cell)
    against CRefCG: fused and four-kernel form."""
    from helpers import two_phase_setup
    from fibergen_amd import LSSolver
    from oracle.c_oracle import CRefCG
    import os
    grid, dims, mixing, iters = (96, 200, 256), (1.5, 1.0, 2.0), "laminate", 3
    mats, phis, normals = two_phase_setup(grid, mixing)
    got = {}
    for fused in (1, 0):
        s = LSSolver(*grid, *dims)
        s.set_num_phases(2)
        for p in range(2):
            s.set_phase(p, mats[p][0], mats[p][1], phis[p])
        s.set_normals(normals)
        s.set_options(mixing_rule=mixing, method="cg", tol=0.0, abs_tol=0.0, maxiter=iters, cg_fused=fused)
        assert s.run(E_LOAD) is False and s.iterations == iters
        got[fused] = (s.ref_material, np.array(s.residuals), s.get_field("epsilon"))
        s.close()
    c = CRefCG(grid, dims, mats, phis, normals, mixing, threads=min(16, os.cpu_count() or 1))
    eps, residuals, it = c.run_cg(E_LOAD, *got[1][0], maxiter=iters)
    assert it == iters
    for fused in (1, 0):
        assert np.abs(got[fused][1] - np.array(residuals)).max() < 1e-11
        assert rel_err(got[fused][2], eps) < 1e-10


def test_porous_mode_on_large_noncubic_grid_four_passes():
    """The scalar modes on a mixed-radix non-cubic grid (x = 3 * 64, y = 5 * 32) against the C loop nests."""
    from helpers import sphere_phi
    from fibergen_amd import LSSolver
    from oracle.c_oracle import CRefScalar
    grid, dims = (192, 160, 128), (1.0, 2.0, 1.5)
    phi = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 10.0], [1.0 - phi, phi]
    E = np.array([1.0, -0.5, 0.25])
    s = LSSolver(*grid, *dims)
    s.set_options(mode="porous")
    s.set_num_phases(2)
    for p in range(2):
        s.set_phase(p, mus[p], 0.0, phis[p])
    s.set_options(tol=0.0, abs_tol=0.0, maxiter=4)
    assert s.run(E) is False and s.iterations == 4
    mu_0, _ = s.ref_material
    got = s.get_field("epsilon")
    s.close()
    c = CRefScalar(grid, dims, mus, phis, threads=_threads())
    g = np.zeros((3,) + grid)
    for _ in range(4):
        g = c.basic_scheme(E, g, mu_0)
    assert rel_err(got, g) < 1e-11
