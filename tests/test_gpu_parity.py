"""GPU parity: every stage kernel and the full basic-scheme loop, called through the
C ABI (libfibergen_amd.so), against the NumPy oracle on the same seeded inputs.

Tolerances (float64): element-wise stages are expected bit-exact (the library is built
without FMA contraction and follows the reference's operation order); they are asserted
to 1e-14 relative so that a libm difference of an ulp in the host-side k-tables cannot
flip the test.  FFT-containing quantities: 1e-12 relative max-norm per stage, 1e-9 on
fields after a converged run, 1e-10 on mean stresses / effective moduli (north_star asks
1e-6 on effective moduli).
"""
import json
import os

import numpy as np
import pytest

from helpers import make_gpu_solver, make_oracle, rel_err, two_phase_setup

pytestmark = pytest.mark.gpu

GRIDS = [
    ((16, 16, 16), (1.0, 1.0, 1.0)),      # all axes on the LDS radix path
    ((32, 16, 64), (1.0, 2.0, 0.5)),      # mixed lengths, anisotropic cell
    ((12, 10, 6), (1.0, 1.0, 1.0)),       # generic DFT path on every axis
    ((41, 33, 11), (41.0, 33.0, 11.0)),   # the reference's self-test grid F:27271 (odd, prime)
    ((10, 1, 1), (1.0, 1.0, 1.0)),        # laminate demo grid (demo/elasticity/laminate)
    ((8, 16, 5), (1.0, 1.0, 1.0)),        # odd nz
    ((24, 40, 48), (1.0, 1.5, 2.0)),      # p * 2^k on every axis (3*8, 5*8, nz/2 = 3*8): sub-line kernels + combine sweep
    ((56, 16, 80), (1.0, 1.0, 1.0)),      # 7*8, power of two, nz/2 = 5*8
    ((72, 24, 240), (1.0, 1.0, 1.0)),     # 9*8, 3*8, nz/2 = 15*8
    ((200, 8, 400), (1.0, 1.0, 1.0)),     # 25*8, nz/2 = 25*8
    ((112, 72, 144), (1.0, 1.0, 1.0)),    # 7*16, 9*8, nz/2 = 9*8: single-kernel passes for p = 7, 9
    ((20, 30, 36), (1.0, 1.5, 0.8)),      # 4*5, 2*3*5, nz/2 = 2*3*3: Stockham tile kernels on every axis (fg_fft_smooth.h)
    ((25, 15, 44), (1.0, 1.0, 1.0)),      # 5*5, 3*5, nz/2 = 2*11
    ((100, 12, 50), (1.0, 1.0, 1.0)),     # the decimal sizes: 4*5*5, nz/2 = 5*5
    ((15, 9, 25), (1.0, 1.0, 1.0)),       # odd nz with small factors: the z rows as nz complex points
]
EXACT = {}


def note(key, a, b):
    EXACT[key] = bool(np.array_equal(a, b))


@pytest.fixture(scope="module", autouse=True)
def dump_exactness():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/stage_bit_exactness.json", "w") as f:
        json.dump(EXACT, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("grid,dims", GRIDS)
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_stress_stage(grid, dims, mixing):
    rng = np.random.default_rng(10)
    o = make_oracle(grid, dims, mixing)
    s = make_gpu_solver(grid, dims, mixing, mu_0=0.77, lambda_0=0.31)
    eps = rng.standard_normal((6,) + grid)
    s.set_field("epsilon", eps)
    s.run_stage("stress")
    got = s.get_field("tau")
    ref = o.calc_stress(0.77, 0.31, eps)
    note("stress/%s/%s" % (mixing, "x".join(map(str, grid))), got, ref)
    assert rel_err(got, ref) < 1e-14
    # sigma = C:eps (C0 = 0), get_raw_field('sigma') F:15496-15508
    assert rel_err(s.get_field("sigma"), o.calc_stress(0.0, 0.0, eps)) < 1e-14
    # mean stress (tree-reduced on the GPU): summation order differs => 1e-13
    o.eps = eps
    assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-12
    assert rel_err(s.mean_strain(), o.mean_strain()) < 1e-12 + 1e-13


@pytest.mark.parametrize("grid,dims", GRIDS)
def test_div_and_eps_stages(grid, dims):
    rng = np.random.default_rng(11)
    o = make_oracle(grid, dims)
    s = make_gpu_solver(grid, dims)
    tau = rng.standard_normal((6,) + grid)
    s.set_field("tau", tau)
    s.run_stage("div")
    got = s.get_field("f")
    ref = o.div_staggered(tau)
    note("div/" + "x".join(map(str, grid)), got, ref)
    assert rel_err(got, ref) < 1e-14
    u = rng.standard_normal((3,) + grid)
    E = np.array([0.3, -0.2, 0.1, 0.05, -0.07, 0.02])
    s.set_field("u", u)
    s.run_stage("eps", E)
    got = s.get_field("epsilon")
    ref = o.eps_staggered(E, u)
    note("eps/" + "x".join(map(str, grid)), got, ref)
    assert rel_err(got, ref) < 1e-14
    # fused sums of squares feeding component_norm F:10127
    assert rel_err(s.get_field("sumsq"), (ref.reshape(6, -1) ** 2).sum(axis=1)) < 1e-12


@pytest.mark.parametrize("grid,dims", GRIDS + [((64, 64, 64), (1.0, 1.0, 1.0)), ((128, 8, 256), (1.0, 1.0, 1.0)),
                                       # every line length of the z passes with the wave-shuffle mirror: nz/2 = 64 ... 512
                                       # (T = 8, 16, 32, 64 lanes per line), and nz/2 = 1024 (two loads / LDS split)
                                       ((8, 8, 128), (1.0, 1.0, 1.0)), ((8, 16, 512), (1.0, 1.0, 1.0)),
                                       ((8, 8, 1024), (1.0, 1.0, 1.0)), ((4, 8, 2048), (1.0, 1.0, 1.0)),
                                       ((512, 8, 16), (1.0, 1.0, 1.0)), ((8, 1024, 16), (1.0, 1.0, 1.0)),
                                       # Stockham tile kernels (fg_fft_smooth.h): the decimal sizes; 8- and 4-column tiles (> 600
                                       # points); nz / 2 = 500 (four rows per tile), 7 * 11 * 13
                                       ((100, 200, 300), (1.0, 1.0, 1.0)), ((500, 12, 400), (1.0, 1.0, 1.0)),
                                       ((12, 1000, 20), (1.0, 1.0, 1.0)), ((1001, 10, 1000), (1.0, 1.0, 1.0)),
                                       ((120, 240, 60), (1.0, 1.0, 1.0)), ((6, 10, 2002), (1.0, 1.0, 1.0)),
                                       # p * 2^k lengths >= 100 on the tile kernels (x = 384: sub-line kernels for x, tile kernels for y / z)
                                       ((384, 144, 160), (1.0, 1.0, 1.0)), ((192, 384, 224), (1.0, 1.0, 1.0)), ((320, 112, 768), (1.0, 1.0, 1.0)),
                                       # odd nz: the rows as nz complex points through the same passes
                                       ((75, 45, 125), (1.0, 1.0, 1.0)), ((12, 10, 225), (1.0, 1.0, 1.0)), ((9, 6, 1001), (1.0, 1.0, 1.0))])
def test_fft_forward_inverse(grid, dims):
    rng = np.random.default_rng(12)
    o = make_oracle(grid, dims) if max(grid) <= 64 else None
    s = make_gpu_solver(grid, dims)
    f = rng.standard_normal((3,) + grid)
    s.set_field("f", f)
    s.run_stage("fft_forward")
    got = s.get_field("f_hat")
    ref = np.fft.rfftn(f, axes=(1, 2, 3)) / float(np.prod(grid))
    assert rel_err(got, ref) < 1e-13
    # inverse of an arbitrary (non-Hermitian) spectrum: FFTW c2r semantics
    nzc = grid[2] // 2 + 1
    spec = rng.standard_normal((3,) + grid[:2] + (nzc,)) + 1j * rng.standard_normal((3,) + grid[:2] + (nzc,))
    s.set_field("f_hat", spec)
    s.run_stage("fft_inverse")
    got = s.get_field("f")  # raw work buffer ("u" would trigger the displacement reconstruction)
    ref = np.fft.irfftn(spec, s=grid, axes=(1, 2, 3)) * float(np.prod(grid))
    assert rel_err(got, ref) < 1e-13


@pytest.mark.parametrize("seed", range(24))
def test_fft_random_grids_with_small_prime_factors(seed):
    """Random grids whose lengths have prime factors <= 13 (even and odd nz): whatever mixture of kernels the plan picks per
    axis -- power-of-two, p * 2^k, the Stockham tile kernels with their planner's radices, tile widths and rows per tile --
    the transform equals numpy's to 1e-13, forward and (for a non-Hermitian spectrum) inverse."""
    rng = np.random.default_rng(1000 + seed)
    smooth = [n for n in range(2, 161) if all(n % p for p in (17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89,
                                                               97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157))]
    grid = tuple(int(rng.choice(smooth)) for _ in range(3))
    s = make_gpu_solver(grid)
    f = rng.standard_normal((3,) + grid)
    s.set_field("f", f)
    s.run_stage("fft_forward")
    got = s.get_field("f_hat")
    ref = np.fft.rfftn(f, axes=(1, 2, 3)) / float(np.prod(grid))
    assert rel_err(got, ref) < 1e-13, grid
    nzc = grid[2] // 2 + 1
    spec = rng.standard_normal((3,) + grid[:2] + (nzc,)) + 1j * rng.standard_normal((3,) + grid[:2] + (nzc,))
    s.set_field("f_hat", spec)
    s.run_stage("fft_inverse")
    got = s.get_field("f")
    ref = np.fft.irfftn(spec, s=grid, axes=(1, 2, 3)) * float(np.prod(grid))
    assert rel_err(got, ref) < 1e-13, grid
    # ... and one pass of the loop through the Green operator (fused x pass where the plan has one) against the oracle
    if max(grid) <= 64:
        o = make_oracle(grid)
        E = np.array([1.0, 0, 0, 0, 0, 0.5])
        s.calc_ref_material()
        o.calc_ref_material()
        eps0 = rng.standard_normal((6,) + grid)
        s.set_field("epsilon", eps0)
        s.iterate(E, 1)
        assert rel_err(s.get_field("epsilon"), o.basic_scheme(E, eps0)) < 1e-11, grid
    s.close()


@pytest.mark.parametrize("grid,dims", GRIDS)
def test_green_operator_stage(grid, dims):
    rng = np.random.default_rng(13)
    o = make_oracle(grid, dims)
    s = make_gpu_solver(grid, dims, mu_0=1324.3, lambda_0=324.2)  # F:24007-24008
    nzc = grid[2] // 2 + 1
    spec = rng.standard_normal((3,) + grid[:2] + (nzc,)) + 1j * rng.standard_normal((3,) + grid[:2] + (nzc,))
    for alpha in (-1.0, 1.0):
        s.set_field("f_hat", spec)
        s.run_stage("g0", np.array([alpha, 0, 0, 0, 0, 0.0]))
        got = s.get_field("f_hat")
        ref = o.g0_apply(1324.3, 324.2, spec, alpha)
        mask = np.isfinite(ref)  # ny = nz = 1 grids: only the (0,0,0) bin is 0/0 before being zeroed
        assert mask.all()
        note("g0/%g/%s" % (alpha, "x".join(map(str, grid))), got, ref)
        assert rel_err(got, ref) < 1e-14


@pytest.mark.parametrize("grid,dims", GRIDS)
def test_staggered_epsG0div_identity_on_gpu(grid, dims):
    """The reference's own self test F:24129-24151 run through the HIP stages."""
    rng = np.random.default_rng(14)
    s = make_gpu_solver(grid, dims, mu_0=1324.3, lambda_0=324.2)
    u0 = rng.standard_normal((3,) + grid)
    s.set_field("u", u0)
    s.run_stage("eps", np.zeros(6))
    e_org = s.get_field("epsilon")
    s.run_stage("stress_const")
    s.run_stage("div")
    s.run_stage("fft_forward")
    s.run_stage("g0", np.array([1.0, 0, 0, 0, 0, 0]))
    s.run_stage("fft_inverse")
    s.run_stage("eps", np.zeros(6))
    e = s.get_field("epsilon")
    diff = np.abs(e - e_org).reshape(6, -1).max(axis=1)
    assert np.linalg.norm(diff) <= np.sqrt(np.finfo(float).eps) * max(1.0, np.abs(e_org).max())


@pytest.mark.parametrize("grid,dims", GRIDS)
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_one_iteration(grid, dims, mixing):
    rng = np.random.default_rng(15)
    o = make_oracle(grid, dims, mixing)
    s = make_gpu_solver(grid, dims, mixing)
    o.calc_ref_material()
    mu0, lam0 = s.calc_ref_material()
    assert mu0 == pytest.approx(o.mu_0, rel=1e-15)
    assert lam0 == 0.0
    eps = 0.1 * rng.standard_normal((6,) + grid)
    E = np.array([1.0, 0.2, -0.3, 0.1, 0.0, 0.4])
    s.set_field("epsilon", eps)
    s.run_stage("iteration", E)
    got = s.get_field("epsilon")
    ref = o.basic_scheme(E, eps)
    assert rel_err(got, ref) < 1e-12


@pytest.mark.parametrize("grid", [(16, 16, 16), (12, 10, 6), (32, 32, 32)])
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_full_run_matches_oracle(grid, mixing):
    tol = 1e-8
    o = make_oracle(grid, mixing=mixing, tol=tol)
    s = make_gpu_solver(grid, mixing=mixing, tol=tol)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    assert o.run(E) is False
    assert s.run(E) is False
    assert s.iterations == o.iterations
    r_gpu, r_ref = np.array(s.residuals), np.array(o.residuals)
    assert len(r_gpu) == len(r_ref)
    # residuals are differences of norms: compare with an absolute floor
    assert np.abs(r_gpu - r_ref).max() < 1e-11
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-9
    assert rel_err(s.get_field("sigma"), o.get_field("sigma")) < 1e-9
    assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-10
    assert rel_err(s.get_field("u"), o.get_field("u")) < 1e-8
    assert s.solve_time > 0


def test_homogeneous_and_callbacks():
    grid = (16, 8, 32)
    from fibergen_amd import LSSolver
    s = LSSolver(*grid)
    s.set_num_phases(1)
    s.set_phase(0, 1.3, 0.9, np.ones(grid))
    s.set_options(tol=1e-10)
    E = np.array([1.0, 0.5, -0.2, 0.1, 0.3, -0.4])
    assert s.run(E) is False
    assert np.abs(s.get_field("epsilon") - E[:, None, None, None]).max() < 1e-13
    assert s.residuals[0] == pytest.approx(1.0, abs=1e-14)
    # convergence callback stops the loop after the first iteration (F:21215)
    s2 = make_gpu_solver((16, 16, 16), tol=1e-12)
    calls = []
    s2.set_convergence_callback(lambda: calls.append(1) or True)
    assert s2.run(E) is False
    assert len(calls) == 1 and s2.iterations == 1 and len(s2.residuals) == 1
    # maxiter
    s2.set_convergence_callback(None)
    s2.set_options(maxiter=3)
    assert s2.run(E) is False
    assert s2.iterations == 3 and len(s2.residuals) == 3


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
@pytest.mark.parametrize("grid", [(16, 16, 16), (8, 14, 128)])   # the second grid takes the tiled sweeps (displacement loop)
def test_mixed_bc_uniaxial_stress(grid, mixing):
    """setBCProjector / calcBCMean / applyBCProjector F:20599-20665 on the GPU path."""
    o = make_oracle(grid, mixing=mixing, tol=1e-9, bc_tol=1e-8, maxiter=400)
    s = make_gpu_solver(grid, mixing=mixing, tol=1e-9, bc_tol=1e-8, maxiter=400)
    P = np.zeros((6, 6))
    P[0, 0] = 1.0
    E = np.array([0.01, 0, 0, 0, 0, 0])
    assert o.run(E, S0=np.zeros(6), P=P) is False
    s.set_bc_projector(P)
    assert s.run(E, np.zeros(6)) is False
    assert s.iterations == o.iterations
    assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-9
    assert np.abs(s.mean_stress()[1:]).max() < 1e-7
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8


def test_error_paths():
    from fibergen_amd import LSSolver
    s = LSSolver(8, 8, 8)
    with pytest.raises(RuntimeError, match="No materials"):
        s.run(np.zeros(6))
    with pytest.raises(RuntimeError):
        s.set_num_phases(9)
    s.set_num_phases(3)
    third = np.full((8, 8, 8), 1.0 / 3.0)
    for p in range(3):
        s.set_phase(p, 1.0 + p, 1.0, third)
    s.set_normals(np.zeros((3, 8, 8, 8)))
    s.set_options(mixing_rule="laminate")
    with pytest.raises(RuntimeError, match="two phase"):
        s.run(np.array([1.0, 0, 0, 0, 0, 0]))
    with pytest.raises(RuntimeError, match="projector"):
        s.set_bc_projector(np.full((6, 6), 0.3))
    with pytest.raises(RuntimeError, match="Unknown field"):
        s.get_field("nonsense")
    # NaN in the solution => run() reports failure (F:21202-21208)
    s2 = make_gpu_solver((8, 8, 8))
    s2.set_phase(1, float("nan"), 1.0)
    assert s2.run(np.array([1.0, 0, 0, 0, 0, 0])) is True


@pytest.mark.parametrize("grid,dims", [((16, 16, 16), (1.0, 1.0, 1.0)), ((32, 16, 64), (1.0, 2.0, 0.5)),
                                       ((12, 10, 6), (1.0, 1.0, 1.0)), ((8, 16, 5), (1.0, 1.0, 1.0)),
                                       ((64, 64, 64), (1.0, 1.0, 1.0)), ((128, 16, 128), (1.0, 1.0, 1.0))])
def test_fused_kernels_match_unfused_pipeline(grid, dims):
    """The fused kernels (polarisation+divergence; x-FFT + Green operator + x-FFT^-1) against the
    one-kernel-per-reference-routine pipeline: the exact-order stress/div fusion (u_loop <= 1) must be bit-identical, the
    FFT fusion agrees to FFT rounding, and so does the default (u_loop = 2: effective moduli + FMA, LDS-tiled on the
    last grid)."""
    rng = np.random.default_rng(21)
    eps = 0.1 * rng.standard_normal((6,) + grid)
    E = np.array([1.0, 0.2, -0.3, 0.1, 0.0, 0.4])
    out = {}
    for name, opts in (("plain", dict(fuse_stress_div=0, fuse_x=0, u_loop=1)), ("sd", dict(fuse_stress_div=1, fuse_x=0, u_loop=1)),
                       ("x", dict(fuse_stress_div=0, fuse_x=1, u_loop=1)), ("both", dict(fuse_stress_div=1, fuse_x=1, u_loop=1)),
                       ("fast", dict())):
        s = make_gpu_solver(grid, dims, mu_0=0.9, lambda_0=0.2)
        for k, v in opts.items():
            s._check(s._lib.fg_set_option_i(s._h, k.encode(), v))
        s.set_field("epsilon", eps)
        s.run_stage("iteration", E)
        out[name] = s.get_field("epsilon")
    assert np.array_equal(out["sd"], out["plain"])
    assert np.array_equal(out["both"], out["x"])
    assert rel_err(out["x"], out["plain"]) < 1e-12
    assert rel_err(out["fast"], out["plain"]) < 1e-12
    o = make_oracle(grid, dims)
    o.mu_0, o.lambda_0 = 0.9, 0.2
    assert rel_err(out["both"], o.basic_scheme(E, eps)) < 1e-12


@pytest.mark.parametrize("grid", [(16, 16, 16), (12, 10, 6), (32, 16, 64), (8, 16, 5)])
def test_displacement_based_loop_is_bit_identical(grid):
    """The displacement-based loop (strain never stored inside the loop, one sweep u_k -> norms of
    eps_k and f_{k+1}) against the strain-based loop of the reference: same iteration count, identical
    residual history and bit-identical converged fields."""
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    res = {}
    for flag in (0, 1, 2):
        s = make_gpu_solver(grid, tol=1e-8)
        s._check(s._lib.fg_set_option_i(s._h, b"u_loop", flag))
        assert s.run(E) is False
        res[flag] = (s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.get_field("sigma"), s.mean_stress())
        # a second load case on the same solver object, then raw passes
        assert s.run(np.array([0, 0, 1.0, 0.3, 0, 0])) is False
        s.iterate(E, 3)
        res[flag] += (s.iterations, s.get_field("epsilon"))
    a, b, c = res[0], res[1], res[2]
    # u_loop = 1: the exact-operation-order sweep is bit-identical to the strain-based loop
    assert a[0] == b[0] and a[5] == b[5]
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert np.array_equal(a[6], b[6])
    # u_loop = 2 (default): precomputed effective moduli + FMA, equal to rounding
    assert a[0] == c[0] and a[5] == c[5]
    assert np.abs(a[1] - c[1]).max() < 1e-12
    assert rel_err(c[2], a[2]) < 1e-11 and rel_err(c[3], a[3]) < 1e-11 and rel_err(c[4], a[4]) < 1e-12
    assert rel_err(c[6], a[6]) < 1e-11
    o = make_oracle(grid, tol=1e-8)
    assert o.run(E) is False
    assert o.iterations == c[0] and rel_err(c[2], o.eps) < 1e-9


def test_callback_field_access_inside_displacement_loop():
    """Accessors called from the convergence callback materialise the strain on demand."""
    s = make_gpu_solver((16, 16, 16), tol=1e-8)
    o = make_oracle((16, 16, 16), tol=1e-8)
    seen = []

    def cb():
        seen.append((s.mean_stress().copy(), s.get_field("u").copy()))
        return False
    s.set_convergence_callback(cb)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    assert s.run(E) is False and o.run(E) is False
    assert s.iterations == o.iterations and len(seen) == o.iterations
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-9
    assert rel_err(seen[-1][0], o.mean_stress()) < 1e-10


@pytest.mark.parametrize("grid", [(16, 16, 16), (12, 10, 6)])
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_cg_solver_matches_oracle(grid, mixing):
    """method=cg (runCGElasticity F:23153-23247, the reference's default): same iteration count,
    residual history and fields as the oracle's restatement; and the same fixed point as the basic scheme."""
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    o = make_oracle(grid, mixing=mixing, tol=1e-10)
    s = make_gpu_solver(grid, mixing=mixing, tol=1e-10, method="cg")
    assert o.run_cg(E) is False
    assert s.run(E) is False
    assert s.iterations == o.iterations and len(s.residuals) == len(o.residuals)
    assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-10
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8
    assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-9
    b = make_gpu_solver(grid, mixing=mixing, tol=1e-12)
    assert b.run(E) is False
    assert s.iterations < b.iterations / 3
    assert rel_err(s.mean_stress(), b.mean_stress()) < 1e-5


def test_cg_mixed_bc_and_maxiter():
    grid = (16, 16, 16)
    o = make_oracle(grid, tol=1e-9, bc_tol=1e-8, maxiter=400)
    s = make_gpu_solver(grid, tol=1e-9, bc_tol=1e-8, maxiter=400, method="cg")
    P = np.zeros((6, 6))
    P[0, 0] = 1.0
    E = np.array([0.01, 0, 0, 0, 0, 0])
    assert o.run_cg(E, S0=np.zeros(6), P=P) is False
    s.set_bc_projector(P)
    assert s.run(E, np.zeros(6)) is False
    assert s.iterations == o.iterations
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-7
    s.set_options(maxiter=2)
    assert s.run(E, np.zeros(6)) is False
    assert s.iterations == 2 and len(s.residuals) == 3


# ------------------------------------------------------------------------------------------------
# gamma_scheme = collocated (GammaOperatorCollocated  F:20302-20310), SURVEY 8a row "Gamma c"
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grid,dims", [((9, 7, 5), (1.0, 1.0, 1.0)), ((16, 16, 16), (1.0, 1.0, 1.0)),
                                       ((12, 10, 6), (2.0, 1.0, 0.5)), ((32, 16, 64), (1.0, 1.0, 1.0))])
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_collocated_pass_and_run_match_oracle(grid, dims, mixing):
    E = np.array([1.0, 0.2, -0.3, 0.1, 0.0, 0.5])
    rng = np.random.default_rng(11)
    eps0 = rng.standard_normal((6,) + grid)
    s = make_gpu_solver(grid, dims, mixing, tol=1e-8, gamma_scheme="collocated", mu_0=0.9, lambda_0=0.2)
    o = make_oracle(grid, dims, mixing, tol=1e-8, gamma_scheme="collocated")
    o.mu_0, o.lambda_0 = 0.9, 0.2
    s.set_field("epsilon", eps0)
    s.run_stage("iteration", E)
    one = s.get_field("epsilon")
    ref = o.basic_scheme(E, eps0)
    assert rel_err(one, ref) < 1e-12
    np.testing.assert_allclose(one.reshape(6, -1).mean(axis=1), E, atol=1e-12)   # zero frequency = E
    # converged run
    s2 = make_gpu_solver(grid, dims, mixing, tol=1e-8, gamma_scheme="collocated")
    o2 = make_oracle(grid, dims, mixing, tol=1e-8, gamma_scheme="collocated")
    assert s2.run(E) is False and o2.run(E) is False
    assert s2.iterations == o2.iterations
    np.testing.assert_allclose(s2.residuals, o2.residuals, rtol=0, atol=1e-11)
    assert rel_err(s2.get_field("epsilon"), o2.eps) < 1e-9
    assert rel_err(s2.mean_stress(), o2.mean_stress()) < 1e-10
    s.close()
    s2.close()


def test_collocated_cg_and_restrictions():
    grid = (9, 9, 9)
    E = np.array([0.0, 1.0, 0, 0, 0.3, 0])
    s = make_gpu_solver(grid, tol=1e-8, gamma_scheme="collocated", method="cg")
    o = make_oracle(grid, tol=1e-8, gamma_scheme="collocated")
    assert s.run(E) is False and o.run_cg(E) is False
    assert s.iterations == o.iterations
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-9
    s.close()


@pytest.mark.parametrize("method", ["basic", "cg"])
def test_collocated_mixed_boundary_conditions(method):
    """initBCProjector(tau_hat) / applyBCProjector(eta_hat) of GammaOperatorCollocated  F:20302-20310, F:20219-20225,
    F:20272-20279: uniaxial stress (strain 11 prescribed, the other stress components zero)."""
    grid = (12, 10, 9)
    P = np.zeros((6, 6))
    P[0, 0] = 1.0
    E, S = np.array([0.01, 0, 0, 0, 0, 0]), np.zeros(6)
    s = make_gpu_solver(grid, tol=1e-9, bc_tol=1e-8, maxiter=600, gamma_scheme="collocated", method=method)
    s.set_bc_projector(P)
    o = make_oracle(grid, tol=1e-9, bc_tol=1e-8, maxiter=600, gamma_scheme="collocated")
    assert (o.run_cg(E, S, P) if method == "cg" else o.run(E, S, P)) is False
    assert s.run(E, S) is False
    assert s.iterations == o.iterations
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8
    assert np.abs(s.mean_stress()[1:]).max() < 1e-7 and s.mean_strain()[0] == pytest.approx(0.01, rel=1e-10)
    s.close()


@pytest.mark.parametrize("grid", [(16, 16, 16), (12, 10, 6), (32, 16, 64)])
def test_displacement_based_loop_with_laminate_mixing_is_bit_identical(grid):
    """Laminate mixing in the displacement-based loop, exact-order form (u_loop=1): one sweep u_k -> (norms of eps_k,
    tau), then the divergence.  Same operations as the strain-based pipeline, so iterates are bit-identical."""
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    res = {}
    for flag in (0, 1):
        s = make_gpu_solver(grid, mixing="laminate", tol=1e-8)
        s._check(s._lib.fg_set_option_i(s._h, b"u_loop", flag))
        assert s.run(E) is False
        res[flag] = (s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.get_field("sigma"), s.mean_stress())
        s.iterate(E, 2)
        res[flag] += (s.get_field("epsilon"),)
        s.close()
    a, b = res[0], res[1]
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    for i in (2, 3, 4, 5):
        assert np.array_equal(a[i], b[i])
    o = make_oracle(grid, mixing="laminate", tol=1e-8)
    assert o.run(E) is False
    assert o.iterations == b[0] and rel_err(b[2], o.eps) < 1e-9


@pytest.mark.parametrize("grid", [(16, 16, 16), (12, 10, 6), (32, 16, 64), (16, 16, 128), (6, 20, 130), (8, 14, 256), (10, 16, 100), (7, 14, 80)])
def test_laminate_mixing_as_correction_of_the_voigt_sweep(grid):
    """u_loop=2 with laminate mixing: f = div tau_voigt (the tiled / fast sweep over all voxels) + div (tau_laminate -
    tau_voigt) gathered at the voxels next to an interface.  Same iterates as the strain-state pipeline up to rounding,
    run-to-run reproducible (no atomics), periodic wrap of the correction in all three directions."""
    E = np.array([1.0, -0.2, 0.1, 0.3, 0, 0.5])
    res = {}
    for flag in (0, 2, 2):
        s = make_gpu_solver(grid, mixing="laminate", tol=1e-8)
        s._check(s._lib.fg_set_option_i(s._h, b"u_loop", flag))
        assert s.run(E) is False
        res.setdefault(flag, []).append((s.iterations, np.array(s.residuals), s.get_field("epsilon"),
                                         s.get_field("sigma"), s.mean_stress()))
        s.close()
    a, b, b2 = res[0][0], res[2][0], res[2][1]
    assert a[0] == b[0]
    assert np.abs(a[1] - b[1]).max() < 1e-12
    assert rel_err(b[2], a[2]) < 1e-11 and rel_err(b[3], a[3]) < 1e-11 and rel_err(b[4], a[4]) < 1e-12
    assert np.array_equal(b[1], b2[1]) and np.array_equal(b[2], b2[2])      # reproducible
    o = make_oracle(grid, mixing="laminate", tol=1e-8)
    assert o.run(E) is False
    assert o.iterations == b[0] and rel_err(b[2], o.eps) < 1e-9 and rel_err(b[3], o.get_field("sigma")) < 1e-9


@pytest.mark.parametrize("grid", [(16, 16, 128), (8, 16, 124), (32, 32, 256), (5, 14, 128), (6, 20, 130), (40, 30, 128),
                                  # rows shorter than a tile (nz / 2 = 40 ... 61: one tile per row, its surplus lanes wrap around)
                                  (16, 16, 100), (9, 14, 80), (12, 20, 96), (7, 15, 122), (20, 14, 82)])
def test_tiled_displacement_sweep(grid):
    """u_tile: the LDS-tiled marching variant of the fast sweep (each strain / polarisation value computed once;
    halo rows and lanes, overlapping last tiles, periodic wrap in all directions; from nz / 2 = 40 on) gives the same iterates."""
    E = np.array([0.2, -0.1, 1.0, 0.3, 0, 0.5])
    res = {}
    for flag in (0, 1):
        s = make_gpu_solver(grid, tol=1e-8)
        s.set_options(u_tile=flag)
        assert s.run(E) is False
        res[flag] = (s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress())
        s.close()
    a, b = res[0], res[1]
    assert a[0] == b[0]
    assert np.abs(a[1] - b[1]).max() < 1e-12
    assert rel_err(b[2], a[2]) < 1e-11 and rel_err(b[3], a[3]) < 1e-12


@pytest.mark.parametrize("grid", [(16, 16, 16), (8, 14, 128)])
def test_three_phase_voigt_all_loop_variants(grid):
    """Three phases (nested spheres: core, coating, matrix) with Voigt mixing: strain-state pipeline, exact
    displacement loop and the fast (effective moduli; tiled where nz/2 >= 40) loop against the oracle."""
    from fibergen_amd import LSSolver
    from helpers import sphere_phi
    from oracle.ls_oracle import LSOracle
    core = sphere_phi(grid, 0.2)
    both = sphere_phi(grid, 0.35)
    phis = [1.0 - both, both - core, core]
    mats = [(0.4, 0.6), (2.0, 1.0), (6.0, 8.0)]
    E = np.array([0.5, -0.25, 1.0, 0.1, 0.2, 0.3])
    o = LSOracle(*grid, mats=mats, phis=phis, tol=1e-8)
    assert o.run(E) is False
    for u_loop in (0, 1, 2):
        s = LSSolver(*grid)
        s.set_num_phases(3)
        for p in range(3):
            s.set_phase(p, mats[p][0], mats[p][1], phis[p])
        s.set_options(tol=1e-8, u_loop=u_loop)
        assert s.run(E) is False
        assert s.iterations == o.iterations
        np.testing.assert_allclose(s.residuals, o.residuals, rtol=0, atol=1e-11)
        assert rel_err(s.get_field("epsilon"), o.eps) < 1e-9
        assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-10
        assert s.volume_fraction(2) == pytest.approx(core.mean(), rel=1e-13)
        s.close()


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
@pytest.mark.parametrize("grid", [(16, 16, 16), (12, 10, 6), (8, 14, 128)])
def test_cg_in_displacement_space(grid, mixing):
    """method=cg with u_loop=2 (default) carries the CG vectors as displacements; u_loop=0 keeps the strain vectors of
    runCGElasticity.  Same iteration counts and residual histories (to rounding) as each other and as the oracle;
    accessors called from the convergence callback see the current iterate."""
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    o = make_oracle(grid, mixing=mixing, tol=1e-10)
    assert o.run_cg(E) is False
    out = {}
    for flag in (0, 2):
        s = make_gpu_solver(grid, mixing=mixing, tol=1e-10, method="cg", u_loop=flag)
        seen = []

        def cb():
            seen.append((s.mean_stress().copy(), s.get_field("u").copy(), s.get_field("epsilon").copy()))
            return False
        s.set_convergence_callback(cb)
        assert s.run(E) is False
        out[flag] = (s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress(), s.get_field("u"), seen)
        s.close()
    for flag in (0, 2):
        it, res, eps, sig, u, seen = out[flag]
        assert it == o.iterations and len(res) == len(o.residuals) == len(seen)
        assert np.abs(res - np.array(o.residuals)).max() < 1e-10
        assert rel_err(eps, o.eps) < 1e-8 and rel_err(sig, o.mean_stress()) < 1e-9
        assert rel_err(seen[-1][2], eps) < 1e-12 and rel_err(seen[-1][0], sig) < 1e-12
        assert np.abs(seen[-1][1] - u).max() < 1e-12 * max(1.0, np.abs(u).max())
    assert rel_err(out[2][2], out[0][2]) < 1e-9
    assert np.abs(out[2][4] - out[0][4]).max() < 1e-9 * max(1.0, np.abs(out[0][4]).max())


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
@pytest.mark.parametrize("grid", [(16, 16, 128), (32, 8, 128), (8, 64, 124), (256, 16, 128)])
def test_x_contiguous_intermediate_layout(grid, mixing):
    """x_layout = 1: the forward y pass writes and the inverse y pass reads [zc/8][y][x][8], the fused x pass runs on
    contiguous tiles (the default on large grids).  Same arithmetic as the plain layout: identical results."""
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    out = {}
    for xl in (0, 1):
        # (plane_fft = 0: the plain-layout run must take the very kernels the x-layout run takes, not the plane kernels)
        s = make_gpu_solver(grid, mixing=mixing, tol=1e-8, x_layout=xl, plane_fft=0)
        assert s.run(E) is False
        out[xl] = (s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress())
        s.close()
    assert out[1][0] == out[0][0]
    assert np.array_equal(out[1][1], out[0][1])
    assert np.array_equal(out[1][2], out[0][2])
    o = make_oracle(grid, mixing=mixing, tol=1e-8)
    assert o.run(E) is False
    assert out[1][0] == o.iterations and rel_err(out[1][2], o.eps) < 1e-9
    # ... and in the displacement-space CG
    s = make_gpu_solver(grid, mixing=mixing, tol=1e-9, method="cg", x_layout=1)
    o = make_oracle(grid, mixing=mixing, tol=1e-9)
    assert s.run(E) is False and o.run_cg(E) is False
    assert s.iterations == o.iterations and rel_err(s.get_field("epsilon"), o.eps) < 1e-8
    s.close()
    # ... in the strain-state pipeline (the polarisation field is the scratch the spectrum passes through) with mixed BC
    if grid[0] <= 32:
        P = np.zeros((6, 6))
        P[0, 0] = 1.0
        s = make_gpu_solver(grid, mixing=mixing, tol=1e-9, bc_tol=1e-8, maxiter=400, u_loop=0, x_layout=1)
        s.set_bc_projector(P)
        o = make_oracle(grid, mixing=mixing, tol=1e-9, bc_tol=1e-8, maxiter=400)
        assert o.run([0.01, 0, 0, 0, 0, 0], S0=np.zeros(6), P=P) is False and s.run([0.01, 0, 0, 0, 0, 0], np.zeros(6)) is False
        assert s.iterations == o.iterations and rel_err(s.get_field("epsilon"), o.eps) < 1e-8
        s.close()


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
@pytest.mark.parametrize("grid,xl", [((16, 16, 128), 0), ((16, 16, 128), 1), ((32, 8, 128), 0), ((8, 64, 124), 1), ((64, 16, 64), 0)])
def test_chunked_transform_pairs_are_bit_identical(grid, xl, mixing):
    """pair_chunk = P: the z and y passes run in runs of P x planes, r2c(c) -> y(c) and y^-1(c) -> c2r(c) (on large grids the
    hand-over then stays in the Infinity Cache).  Same kernels on the same lines: every iterate is bit-identical to the
    whole-field passes, with the plain and the x-contiguous intermediate layout, for chunk lengths that do and do not divide
    nx."""
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    out = {}
    for pc in (0, 1, 3, grid[0] // 2, grid[0]):
        s = make_gpu_solver(grid, mixing=mixing, tol=1e-8, x_layout=xl, plane_fft=0, pair_chunk=pc)
        assert s.run(E) is False
        out[pc] = (s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress(), s.get_field("u"))
        s.close()
    for pc in out:
        assert out[pc][0] == out[0][0]
        for k in (1, 2, 3, 4):
            assert np.array_equal(out[pc][k], out[0][k]), (pc, k)
    # ... in the conjugate gradients and the strain-state pipeline
    for kw in (dict(method="cg"), dict(u_loop=0)):
        res = []
        for pc in (0, 3):
            s = make_gpu_solver(grid, mixing=mixing, tol=1e-8, x_layout=xl, plane_fft=0, pair_chunk=pc, **kw)
            assert s.run(E) is False
            res.append((s.iterations, np.array(s.residuals), s.get_field("epsilon")))
            s.close()
        assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("grid", [(16, 16, 16), (12, 10, 6), (8, 16, 5), (32, 8, 128), (40, 24, 66)])
def test_staged_host_transfers_are_byte_identical(grid):
    """staged_copy = 1: fields cross the boundary through the pinned-buffer pipeline (padding stripped / added on the device,
    chunks on the copy engine, a team of host threads on the pageable side) instead of one strided copy -- the same bytes,
    whatever the chunk size (rows that do not divide the chunk, one row per chunk, everything in one chunk)."""
    rng = np.random.default_rng(5)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    mats, phis, normals = two_phase_setup(grid, "laminate")
    ref = None
    for staged, kb in ((0, 16384), (1, 16384), (1, 1), (1, 7), (1, 64)):
        from fibergen_amd import LSSolver
        s = LSSolver(*grid, 1.0, 1.0, 1.0)
        s.set_options(staged_copy=staged, stage_chunk_kb=kb)
        s.set_num_phases(2)
        for p in range(2):
            s.set_phase(p, mats[p][0], mats[p][1], phis[p])
        s.set_normals(normals)
        s.set_options(mixing_rule="laminate", tol=1e-8)
        eps0 = np.random.default_rng(7).standard_normal((6,) + grid)
        s.set_field("epsilon", eps0)
        back = s.get_field("epsilon")
        assert np.array_equal(back, eps0)
        assert np.array_equal(s.get_field("phi")[1], phis[1]) and np.array_equal(s.get_field("normals"), normals)
        assert s.run(E) is False
        got = (s.iterations, s.get_field("epsilon"), s.get_field("sigma"), s.get_field("u"), s.get_field("f_hat"))
        s.close()
        if ref is None:
            ref = got
        else:
            assert got[0] == ref[0]
            for a, b in zip(got[1:], ref[1:]):
                assert np.array_equal(a, b)


@pytest.mark.parametrize("grid,dims", GRIDS[:9] + [((16, 16, 128), (1.0, 1.0, 1.0))])
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
@pytest.mark.parametrize("estimator", ["sigma", "energy"])
def test_sigma_and_energy_estimators_match_oracle(grid, dims, mixing, estimator):
    """error_estimator = sigma (SigmaErrorEstimator F:14514-14587) / energy (EnergyErrorEstimator F:14410-14468): the stop
    rule watches <sigma> / <W> of the strain field after every iteration.  Iteration counts equal the oracle's, residual
    histories to 1e-9 (differences of means that agree to rounding, divided by the mean), basic scheme and conjugate gradients;
    the last grid runs the displacement loop with its tiled sweep (the strain field is materialised for the measurement)."""
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    tol = 1e-6
    for method in ("basic", "cg"):
        o = make_oracle(grid, dims, mixing, tol=tol, error_estimator=estimator)
        s = make_gpu_solver(grid, dims, mixing, tol=tol, error_estimator=estimator, method=method)
        assert (o.run_cg(E) if method == "cg" else o.run(E)) is False
        assert s.run(E) is False
        assert s.iterations == o.iterations and len(s.residuals) == len(o.residuals), (method, s.iterations, o.iterations)
        assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-9
        assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8 and rel_err(s.mean_stress(), o.mean_stress()) < 1e-9
        s.close()


@pytest.mark.parametrize("method", ["basic", "cg"])
def test_none_estimator_runs_to_maxiter(method):
    """error_estimator = none (NoneErrorEstimator F:14370-14378): abs = rel = 1, the run ends at maxiter (or on a callback's
    request); residual with the basic scheme is the reference's run-time error (ErrorEstimator::update F:14353)."""
    grid = (16, 16, 16)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    o = make_oracle(grid, tol=1e-6, error_estimator="none", maxiter=9)
    s = make_gpu_solver(grid, tol=1e-6, error_estimator="none", maxiter=9, method=method)
    assert (o.run_cg(E) if method == "cg" else o.run(E)) is False and s.run(E) is False
    assert s.iterations == o.iterations == 9 and list(s.residuals) == o.residuals and set(o.residuals) == {1.0}
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-10
    calls = []
    s.set_convergence_callback(lambda: calls.append(1) or len(calls) >= 3)
    assert s.run(E) is False and len(calls) == 3
    s.close()
    with pytest.raises(RuntimeError, match="Unknown error estimator"):
        make_gpu_solver(grid, error_estimator="div_sigma")
    s = make_gpu_solver(grid, mixing="voigt")
    s.set_options(mode="porous", error_estimator="sigma")
    with pytest.raises(RuntimeError, match="heat / porous mode supports the error estimators"):
        s.run(np.array([1.0, 0, 0]))
    s.close()


def test_laminate_rule_at_oblique_normals_is_the_rotated_closed_form():
    """The HIP laminate rule (get_field('sigma'), FG_STAGE_STRESS) at random oblique normals against the reference-held
    closed form: P(eps, n) = R [C_lam : (R^T eps R)] R^T with C_lam from calc_isotropic_laminate F:26412-26446 and R e_x = n
    (tests/test_oracle_pins.py holds the same statement for the oracle)."""
    from fibergen_amd import LSSolver
    from helpers import INCLUSION, MATRIX, lame
    from test_oracle_pins import (_apply_stiffness, _reference_laminate_formula, _rotation_taking_ex_to, _to_matrix,
                                  _to_vector)
    rng = np.random.default_rng(21)
    grid = (8, 6, 10)
    nv = grid[0] * grid[1] * grid[2]
    mats = [lame(**MATRIX), lame(**INCLUSION)]
    n = rng.standard_normal((3, nv))
    n /= np.linalg.norm(n, axis=0)
    c1 = rng.uniform(0.02, 0.98, nv)
    eps = rng.standard_normal((6, nv))
    s = LSSolver(*grid)
    s.set_num_phases(2)
    s.set_phase(0, *mats[0], c1.reshape(grid))
    s.set_phase(1, *mats[1], (1 - c1).reshape(grid))
    s.set_normals(n.reshape((3,) + grid))
    s.set_options(mixing_rule="laminate")
    s.set_field("epsilon", eps.reshape((6,) + grid))
    got = s.get_field("sigma").reshape(6, nv)
    s.close()
    for v in range(nv):
        R = _rotation_taking_ex_to(n[:, v])
        C = _reference_laminate_formula([(c1[v], *mats[0]), (1 - c1[v], *mats[1])])
        want = _to_vector(R @ _to_matrix(_apply_stiffness(C, _to_vector(R.T @ _to_matrix(eps[:, v]) @ R))) @ R.T)
        assert np.abs(got[:, v] - want).max() < 1e-11 * np.abs(want).max()


def test_stress_driven_laminate_demo_closed_form():
    """The mixed-BC projector on the GPU against the laminate closed form (F:26412-26446): the three-layer medium of
    demo/elasticity/laminate under a fully prescribed mean stress and under uniaxial strain with free lateral faces."""
    from fibergen_amd import LSSolver
    from oracle.ls_oracle import material_from_pair
    from test_oracle_pins import _reference_laminate_formula
    shape = (10, 1, 1)
    mats = [material_from_pair(E=100.0, nu=0.4), material_from_pair(E=25.0, nu=0.25), material_from_pair(E=50.0, nu=0.3)]
    fr = [0.2, 0.3, 0.5]
    edges = np.round(np.cumsum([0.0] + fr) * 10).astype(int)
    Cw = _reference_laminate_formula([(f, m["mu"], m["lambda"]) for f, m in zip(fr, mats)]) * np.array([1, 1, 1, 2, 2, 2.0])[None, :]

    def solver():
        s = LSSolver(*shape)
        s.set_num_phases(3)
        for p, (a, b) in enumerate(zip(edges[:-1], edges[1:])):
            phi = np.zeros(shape)
            phi[a:b] = 1.0
            s.set_phase(p, mats[p]["mu"], mats[p]["lambda"], phi)
        s.set_options(tol=1e-13, bc_tol=1e-10, maxiter=5000)
        return s
    S = np.array([1.0, 0.2, -0.3, 0.1, 0.05, -0.2])
    s = solver()
    s.set_bc_projector(np.zeros((6, 6)))
    assert s.run(np.zeros(6), S) is False
    want = np.linalg.solve(Cw, S)
    assert np.abs(s.mean_strain() - want).max() < 1e-8 * np.abs(want).max() and np.abs(s.mean_stress() - S).max() < 1e-8
    s.close()
    s = solver()
    P = np.zeros((6, 6))
    P[0, 0] = 1.0
    s.set_bc_projector(P)
    assert s.run(np.array([0.01, 0, 0, 0, 0, 0]), np.zeros(6)) is False
    want = np.concatenate([[0.01], np.linalg.solve(Cw[1:, 1:], -Cw[1:, 0] * 0.01)])
    assert np.abs(s.mean_strain() - want).max() < 1e-8 * 0.01
    assert abs(s.mean_stress()[0] - Cw[0] @ want) < 1e-8 * abs(Cw[0] @ want)
    s.close()


@pytest.mark.parametrize("mode,mixing", [("elasticity", "voigt"), ("elasticity", "laminate"), ("porous", "voigt")])
def test_x_lines_of_1024_points_take_the_fused_pass(mode, mixing):
    """nx = 1024: the fused x pass on half-segment (4-column) tiles -- an 8-column tile's exchange buffer would leave 16 waves
    128 registers each.  Four passes of the default loop against the oracle, and against fuse_x = 0 (three separate passes)."""
    grid = (1024, 16, 128)
    E = np.array([0.01, -0.004, 0.002, 0.003, -0.001, 0.002])
    if mode == "porous":
        from fibergen_amd import LSSolver
        from helpers import sphere_phi
        from oracle.scalar_oracle import ScalarOracle
        phi1 = sphere_phi(grid, 0.3)
        mus, phis = [1.0, 9.0], [1 - phi1, phi1]
        o = ScalarOracle(*grid, mus=mus, phis=phis, tol=-1.0, maxiter=4)
        o.abs_tol = -1.0
        o.run(E[:3])
        out = []
        for fuse in (1, 0):
            s = LSSolver(*grid)
            s.set_options(mode="porous")
            s.set_num_phases(2)
            for p in range(2):
                s.set_phase(p, mus[p], 0.0, phis[p])
            s.set_options(tol=-1.0, abs_tol=-1.0, maxiter=4, fuse_x=fuse)
            s.run(E[:3])
            assert s.iterations == o.iterations == 4
            assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-11
            out.append(s.get_field("epsilon"))
            s.close()
        assert rel_err(out[0], o.eps) < 1e-11 and rel_err(out[0], out[1]) < 1e-12
        return
    o = make_oracle(grid, mixing=mixing, tol=-1.0, maxiter=4)
    o.abs_tol = -1.0
    o.run(E)
    out = []
    for fuse in (1, 0):
        s = make_gpu_solver(grid, mixing=mixing, tol=-1.0, abs_tol=-1.0, maxiter=4, fuse_x=fuse)
        s.run(E)
        assert s.iterations == o.iterations == 4
        assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-11
        out.append(s.get_field("epsilon"))
        s.close()
    assert rel_err(out[0], o.eps) < 1e-11 and rel_err(out[0], out[1]) < 1e-12


@pytest.mark.parametrize("grid,mode,mixing", [((8, 32, 32), "elasticity", "voigt"), ((16, 64, 64), "elasticity", "laminate"),
                                              ((8, 128, 128), "elasticity", "voigt"), ((4, 32, 128), "elasticity", "voigt"),
                                              ((4, 128, 32), "elasticity", "laminate"), ((8, 64, 128), "porous", "voigt"),
                                              ((16, 64, 32), "viscosity", "voigt"), ((8, 256, 64), "elasticity", "voigt"),
                                              ((8, 16, 16), "elasticity", "laminate"), ((4, 256, 16), "elasticity", "voigt"),
                                              ((8, 32, 16), "porous", "voigt")])
def test_plane_fft_equals_the_separate_passes(grid, mode, mixing):
    """plane_fft: the z and y transforms of a z-y plane in one kernel (fg_fft_plane.h; the plane lives in LDS) against the
    separate passes, five passes of the loop: the same butterflies in the same order, so the iterates agree to rounding of
    the contraction choices (<= 1e-14); and against the oracle."""
    from fibergen_amd import LSSolver
    from helpers import sphere_phi
    E = np.array([0.01, -0.004, 0.002, 0.003, -0.001, 0.002])
    out = []
    for plane in (1, 0):
        if mode == "elasticity":
            s = make_gpu_solver(grid, (1.0, 1.2, 0.9), mixing=mixing, tol=-1.0, abs_tol=-1.0, maxiter=5, plane_fft=plane)
            Erun = E
        else:
            phi1 = sphere_phi(grid, 0.3)
            s = LSSolver(*grid, 1.0, 1.2, 0.9)
            s.set_options(mode=mode)
            s.set_num_phases(2)
            s.set_phase(0, 1.0, 0.0, 1 - phi1)
            s.set_phase(1, 7.0, 0.0, phi1)
            s.set_options(tol=-1.0, abs_tol=-1.0, maxiter=5, plane_fft=plane)
            Erun = E[:3] if mode == "porous" else E - np.array([E[:3].mean()] * 3 + [0, 0, 0])
        s.run(Erun)
        out.append((np.array(s.residuals), s.get_field("epsilon"), s.stage_times() if False else None))
        s.close()
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-13
    assert rel_err(out[0][1], out[1][1]) < 1e-13
    if mode == "elasticity":
        o = make_oracle(grid, (1.0, 1.2, 0.9), mixing=mixing, tol=-1.0, maxiter=5)
        o.abs_tol = -1.0
        o.run(E)
        assert np.abs(out[0][0] - np.array(o.residuals)).max() < 1e-11
        assert rel_err(out[0][1], o.eps) < 1e-11


def test_counters_report_the_laminate_lists():
    """fg_get_counter: the lengths of the laminate correction's voxel lists (what bench.py prices its kernels with); unknown
    names give -1."""
    from helpers import two_phase_setup
    grid = (16, 16, 128)
    s = make_gpu_solver(grid, mixing="laminate", tol=1e-6)
    assert s.counter("interface_voxels") == 0 and s.counter("no_such_counter") == -1
    assert s.run(np.array([1.0, 0, 0, 0, 0, 0.5])) is False
    _, phis, _ = two_phase_setup(grid, "laminate")
    mixed = int((((phis[0] != 0) & (phis[0] != 1)) | ((phis[1] != 0) & (phis[1] != 1))).sum())   # k_mixed_list's criterion
    assert s.counter("interface_voxels") == mixed > 0
    # every interface voxel is affected, and so are its six stencil neighbours at most
    assert mixed <= s.counter("affected_voxels") <= 7 * mixed
    s.close()


@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
@pytest.mark.parametrize("grid", [(20, 12, 14), (36, 10, 22), (100, 6, 18), (120, 4, 10), (200, 3, 12), (240, 2, 10), (300, 3, 8),
                                  (225, 2, 6), (480, 2, 6), (500, 2, 4)])
def test_tile_kernels_fused_x_pass_joint_and_per_component(grid, mixing):
    """joint_x = 1 (default): the fused x pass of the tile kernels on ONE joint image of the three components (k_smooth_xjoint:
    kernels per largest radix, 256 / 512 threads, 4-column tiles from 420 points on); joint_x = 0: one image per component
    (lines <= 416 points; longer ones: three separate kernels).  Both against the oracle: same iteration counts, residual
    histories and strain fields."""
    E = np.array([1.0, 0, 0, 0.2, 0, 0.5])
    o = make_oracle(grid, mixing=mixing, tol=1e-8)
    assert o.run(E) is False
    for joint in (1, 0):
        s = make_gpu_solver(grid, mixing=mixing, tol=1e-8, joint_x=joint)
        assert s.run(E) is False
        assert s.iterations == o.iterations, (joint, s.iterations, o.iterations)
        np.testing.assert_allclose(s.residuals, o.residuals, rtol=0, atol=1e-10)
        assert rel_err(s.get_field("epsilon"), o.eps) < 1e-10
        assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-11
        s.close()


@pytest.mark.parametrize("grid", [(100, 100, 100), (120, 150, 180), (200, 240, 50), (300, 36, 100), (400, 12, 200), (480, 10, 150),
                                  (500, 10, 120), (150, 30, 400), (180, 20, 480), (250, 40, 500)])
def test_tile_kernels_built_for_one_plan_equal_the_class_kernels(grid):
    """tile_plans = 1 (default): lengths in the tables of fg_fft_smooth_plans.h run kernels built for their plan (line length,
    tile shape and radices as template parameters); tile_plans = 0: the class kernels, which take any plan.  The same
    butterflies in the same order: forward and inverse transforms against numpy <= 1e-13 in both forms, the two forms equal to
    the contraction choices of two compilations, and three passes of the loop (fused x pass) give the same strain field."""
    rng = np.random.default_rng(grid[0] + grid[1])
    f = rng.standard_normal((3,) + grid)
    nzc = grid[2] // 2 + 1
    spec = rng.standard_normal((3,) + grid[:2] + (nzc,)) + 1j * rng.standard_normal((3,) + grid[:2] + (nzc,))
    ref_f = np.fft.rfftn(f, axes=(1, 2, 3)) / float(np.prod(grid))
    ref_i = np.fft.irfftn(spec, s=grid, axes=(1, 2, 3)) * float(np.prod(grid))
    E = np.array([1.0, 0, 0, 0.3, 0, 0.5])
    out = {}
    try:
        for flag in (1, 0):
            s = make_gpu_solver(grid, tile_plans=flag)
            s.set_field("f", f)
            s.run_stage("fft_forward")
            fwd = s.get_field("f_hat")
            assert rel_err(fwd, ref_f) < 1e-13, flag
            s.set_field("f_hat", spec)
            s.run_stage("fft_inverse")
            inv = s.get_field("f")
            assert rel_err(inv, ref_i) < 1e-13, flag
            s.calc_ref_material()
            s.iterate(E, 3)
            out[flag] = (fwd, inv, s.get_field("epsilon"))
            s.close()
    finally:
        s = make_gpu_solver((8, 8, 8), tile_plans=1)   # (the switch is process-wide)
        s.close()
    assert rel_err(out[1][0], out[0][0]) < 1e-14 and rel_err(out[1][1], out[0][1]) < 1e-14
    assert rel_err(out[1][2], out[0][2]) < 1e-12


def _plan_table(name):
    """entries of a plan-kernel table of fg_fft_smooth_plans.h (first field = the line length)"""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fibergen_amd", "csrc",
                            "fg_fft_smooth_plans.h")).read()
    body = src[src.index("#define " + name + "(X)"):]
    body = body[:body.index("\n\n")]
    return sorted({int(m.group(1)) for m in re.finditer(r"X\((\d+),", body)})


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_every_plan_kernel_against_numpy(axis):
    """every length in the plan kernels' tables, as x (strided pass + fused x pass in the loop), y (strided pass) and z (2 M real
    points) of a thin grid: forward and inverse transforms against numpy <= 1e-13 and -- for x -- two passes of the loop with the
    plan kernels on and off"""
    lengths = _plan_table("FG_SMOOTH_Z_PLANS") if axis == 2 else sorted(set(_plan_table("FG_SMOOTH_STRIDED_PLANS")) |
                                                                        set(_plan_table("FG_SMOOTH_X_PLANS")))
    rng = np.random.default_rng(axis)
    E = np.array([1.0, 0, 0, 0.3, 0, 0.5])
    for n in lengths:
        grid = [(n, 3, 10), (5, n, 6), (3, 4, 2 * n)][axis]
        s = make_gpu_solver(grid)
        f = rng.standard_normal((3,) + grid)
        s.set_field("f", f)
        s.run_stage("fft_forward")
        assert rel_err(s.get_field("f_hat"), np.fft.rfftn(f, axes=(1, 2, 3)) / float(np.prod(grid))) < 1e-13, grid
        nzc = grid[2] // 2 + 1
        spec = rng.standard_normal((3,) + grid[:2] + (nzc,)) + 1j * rng.standard_normal((3,) + grid[:2] + (nzc,))
        s.set_field("f_hat", spec)
        s.run_stage("fft_inverse")
        assert rel_err(s.get_field("f"), np.fft.irfftn(spec, s=grid, axes=(1, 2, 3)) * float(np.prod(grid))) < 1e-13, grid
        if axis == 0:
            eps = {}
            try:
                for flag in (1, 0):
                    t = make_gpu_solver(grid, tile_plans=flag)
                    t.calc_ref_material()
                    t.iterate(E, 2)
                    eps[flag] = t.get_field("epsilon")
                    t.close()
            finally:
                s.set_options(tile_plans=1)
            assert rel_err(eps[1], eps[0]) < 1e-12, grid
        s.close()
