"""The C/OpenMP restatement (oracle/c) against the NumPy oracle: two independent
restatements of the same reference routines must agree (bit for bit on the element-wise
stages)."""
import numpy as np
import pytest

from helpers import make_oracle, rel_err, two_phase_setup
from oracle.c_oracle import CRef

GRIDS = [((8, 6, 4), (1.0, 2.0, 3.0)), ((12, 10, 7), (1.0, 1.0, 1.0)), ((16, 16, 16), (1.0, 1.0, 1.0)), ((5, 1, 1), (1, 1, 1))]


@pytest.mark.parametrize("grid,dims", GRIDS)
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_c_stages_equal_numpy_oracle(grid, dims, mixing):
    rng = np.random.default_rng(7)
    o = make_oracle(grid, dims, mixing)
    mats, phis, normals = two_phase_setup(grid, mixing)
    c = CRef(grid, dims, mats, phis, normals, mixing, threads=2)
    eps = rng.standard_normal((6,) + grid)
    assert np.array_equal(c.calc_stress(0.7, 0.3, eps), o.calc_stress(0.7, 0.3, eps))
    tau = rng.standard_normal((6,) + grid)
    assert np.array_equal(c.div(tau), o.div_staggered(tau))
    u = rng.standard_normal((3,) + grid)
    E = np.array([0.3, -0.2, 0.1, 0.05, -0.07, 0.02])
    assert np.array_equal(c.eps_op(E, u), o.eps_staggered(E, u))
    nzc = grid[2] // 2 + 1
    spec = rng.standard_normal((3,) + grid[:2] + (nzc,)) + 1j * rng.standard_normal((3,) + grid[:2] + (nzc,))
    for alpha in (-1.0, 1.0):
        assert rel_err(c.g0(1324.3, 324.2, spec.copy(), alpha), o.g0_apply(1324.3, 324.2, spec, alpha)) < 1e-14
    o.eps = eps
    assert rel_err(c.mean_stress(eps), o.mean_stress()) < 1e-12
    assert rel_err(c.component_norm(eps), o.component_norm(eps)) < 1e-13
    o.calc_ref_material()
    e1 = c.basic_scheme(E, eps, o.mu_0, o.lambda_0)
    e2 = o.basic_scheme(E, eps)
    assert rel_err(e1, e2) < 1e-12


@pytest.mark.parametrize("grid,dims", GRIDS[:3])
def test_c_scalar_and_viscosity_passes_equal_numpy_oracles(grid, dims):
    """config 5's checkers at 256^3 (tests/test_gpu_fullsize_oracle.py) are the C loop nests; here they are held against
    the NumPy restatements of the same routines (oracle/scalar_oracle.py, oracle/viscosity_oracle.py)."""
    from oracle.c_oracle import CRefScalar, CRefViscosity
    from oracle.scalar_oracle import ScalarOracle
    from oracle.viscosity_oracle import ViscosityOracle
    rng = np.random.default_rng(11)
    _, phis, _ = two_phase_setup(grid)
    mus = [1.0, 10.0]
    so = ScalarOracle(*grid, mus=mus, phis=phis, dx=dims[0], dy=dims[1], dz=dims[2])
    so.calc_ref_material()
    cs = CRefScalar(grid, dims, mus, phis, threads=2)
    g = rng.standard_normal((3,) + grid)
    E3 = np.array([0.3, -0.2, 0.1])
    assert rel_err(cs.basic_scheme(E3, g, so.mu_0), so.basic_scheme(E3, g)) < 1e-12
    tau = np.empty_like(g)
    import ctypes
    from oracle.c_oracle import _P
    cs.lib.ref_calc_stress_scalar(*grid, _P(g), _P(cs.phi), 2, _P(cs.mu), ctypes.c_double(0.7), ctypes.c_double(1.0), _P(tau))
    assert np.array_equal(tau, so.calc_stress(0.7, g))
    f = np.empty(grid)
    cs.lib.ref_div_heat(*grid, *map(ctypes.c_double, dims), _P(tau), _P(f))
    assert np.array_equal(f, so.div_heat(tau))
    out = np.empty_like(g)
    T = rng.standard_normal(grid)
    cs.lib.ref_eps_heat(*grid, *map(ctypes.c_double, dims), _P(E3), _P(T), _P(out))
    assert np.array_equal(out, so.eps_heat(E3, T))

    vmus = [1.0, 0.1]
    vo = ViscosityOracle(*grid, *dims, mats=[(m, 0.0) for m in vmus], phis=phis)
    vo.calc_ref_material()
    cv = CRefViscosity(grid, dims, vmus, phis, threads=2)
    eps = rng.standard_normal((6,) + grid)
    E6 = np.array([1.0, -1.0, 0.0, 0.2, -0.1, 0.3])
    assert np.array_equal(cv.calc_stress(vo.mu_0, 0.0, eps), vo.calc_stress(vo.mu_0, vo.lambda_0, eps))
    assert rel_err(cv.basic_scheme(E6, eps, vo.mu_0), vo.basic_scheme(E6, eps)) < 1e-12


@pytest.mark.parametrize("grid,dims", GRIDS[:3])
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
@pytest.mark.parametrize("estimator", ["epsilon", "residual"])
def test_c_cg_equals_numpy_oracle_per_iteration(grid, dims, mixing, estimator):
    """runCGElasticity on the C loop nests (CRefCG: the checker of the GPU's CG at 256^3 / 512^3) against LSOracle._run_cg_step,
    iteration by iteration: the residual history after k iterations and the strain field of every k."""
    from oracle.c_oracle import CRefCG
    E = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.5])
    mats, phis, normals = two_phase_setup(grid, mixing)
    c = CRefCG(grid, dims, mats, phis, normals, mixing, threads=2)
    for k in range(4):
        o = make_oracle(grid, dims, mixing, tol=0.0)
        o.abs_tol = 0.0
        o.maxiter = k
        o.error_estimator = estimator
        assert o.run_cg(E) is False
        eps, res, it = c.run_cg(E, o.mu_0, o.lambda_0, maxiter=k, estimator=estimator)
        assert it == o.iterations == k and len(res) == len(o.residuals) == k + 1
        assert np.abs(np.array(res) - np.array(o.residuals)).max() < 1e-13
        assert rel_err(eps, o.eps) < 1e-12
    # a converged run stops at the same iteration
    o = make_oracle(grid, dims, mixing, tol=1e-5)
    o.error_estimator = estimator
    assert o.run_cg(E) is False
    eps, res, it = c.run_cg(E, o.mu_0, o.lambda_0, maxiter=o.maxiter, tol=1e-5, abs_tol=o.abs_tol, estimator=estimator)
    assert it == o.iterations and rel_err(eps, o.eps) < 1e-11


@pytest.mark.parametrize("grid", [(4, 8, 16), (8, 4, 8), (2, 2, 4), (16, 32, 64), (64, 16, 32)])
def test_c_fft_matches_numpy(grid):
    """oracle/c's own threaded row-column transform (the cpu_baseline's stand-in for threaded FFTW): r2c against numpy's rfftn,
    c2r with FFTW's semantics on a spectrum that is NOT Hermitian (imaginary parts of the DC / Nyquist bins ignored), round trip."""
    import ctypes
    from oracle.c_oracle import _P, load
    lib = load()
    lib.ref_fft_plan.restype = ctypes.c_void_p
    assert lib.ref_fft_supported(*grid) == 1 and lib.ref_fft_supported(6, 8, 8) == 0
    lib.ref_set_threads(3)
    plan = ctypes.c_void_p(lib.ref_fft_plan(*grid))
    rng = np.random.default_rng(5)
    f = rng.standard_normal((3,) + grid)
    fh = np.empty((3,) + grid[:2] + (grid[2] // 2 + 1,), dtype=np.complex128)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.ref_fft_r2c(plan, 3, _P(f), fh.ctypes.data_as(dp))
    ref = np.fft.rfftn(f, axes=(1, 2, 3))
    assert np.abs(fh - ref).max() < 1e-13 * np.abs(ref).max()
    spec = rng.standard_normal(fh.shape) + 1j * rng.standard_normal(fh.shape)
    want = np.fft.irfftn(spec, s=grid, axes=(1, 2, 3)) * float(np.prod(grid))
    u = np.empty_like(f)
    lib.ref_fft_c2r(plan, 3, spec.copy().ctypes.data_as(dp), _P(u))
    assert np.abs(u - want).max() < 1e-13 * np.abs(want).max()
    lib.ref_fft_c2r(plan, 3, fh.ctypes.data_as(dp), _P(u))
    assert np.abs(u / float(np.prod(grid)) - f).max() < 1e-13
    lib.ref_fft_plan_free(plan)


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_cpu_baseline_loop_equals_checker(mixing):
    """CRefLoop (bench.py's cpu_baseline: in place, own transform, placed buffers) performs the passes of CRef (the checker)."""
    from oracle.c_oracle import CRefLoop
    grid, dims = (16, 16, 16), (1.0, 1.0, 1.0)
    mats, phis, normals = two_phase_setup(grid, mixing)
    E = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.5])
    c = CRef(grid, dims, mats, phis, normals, mixing, threads=2)
    loops = [CRefLoop(grid, dims, mats, phis, normals, mixing, threads=2, native=False, fft=f) for f in ("own", "pocketfft")]
    assert loops[0].own_fft and not loops[1].own_fft
    eps = np.zeros((6,) + grid)
    for _ in range(3):
        eps = c.basic_scheme(E, eps, 0.7, 0.0)
        for lp in loops:
            lp.one_pass(E, 0.7, 0.0)
    for lp in loops:
        assert rel_err(lp.eps, eps) < 1e-12
        assert rel_err(lp.norms, c.component_norm(eps)) < 1e-12


@pytest.mark.parametrize("grid,dims", GRIDS)
def test_contiguous_stencil_operators_equal_reference_order(grid, dims):
    """ref_div_contig / ref_eps_contig (z innermost; cpu_baseline's "tuned_loops" figure) against the loop nests in the
    reference's traversal orders: the same values bit for bit."""
    import ctypes
    from oracle.c_oracle import _P, load
    lib = load()
    lib.ref_set_threads(3)
    rng = np.random.default_rng(3)
    d = [ctypes.c_double(v) for v in dims]
    tau = rng.standard_normal((6,) + grid)
    f1, f2 = np.empty((3,) + grid), np.empty((3,) + grid)
    lib.ref_div(*grid, *d, _P(tau), _P(f1))
    lib.ref_div_contig(*grid, *d, _P(tau), _P(f2))
    assert np.array_equal(f1, f2)
    u = rng.standard_normal((3,) + grid)
    E = np.array([0.3, -0.2, 0.1, 0.05, -0.07, 0.02])
    e1, e2 = np.empty((6,) + grid), np.empty((6,) + grid)
    lib.ref_eps(*grid, *d, _P(E), _P(u), _P(e1))
    lib.ref_eps_contig(*grid, *d, _P(E), _P(u), _P(e2))
    assert np.array_equal(e1, e2)
