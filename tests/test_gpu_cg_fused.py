"""Conjugate gradients with the fused tiled vector sweeps (round 4, option cg_fused; runCGElasticity F:23153-23247): the
update of the iterate and the residual with their norms as ONE out-of-place sweep, p:(p - w) as a tiled sweep, the direction
update p = r + beta p formed inside the operator's displacement sweep.  Same arithmetic as the four point-wise / dot kernels
(to FMA rounding): iteration counts and residual histories are the oracle's, and equal to the unfused form's."""
import numpy as np
import pytest

from helpers import make_gpu_solver, make_oracle, rel_err, sphere_phi

pytestmark = pytest.mark.gpu

E_LOAD = [1.0, 0, 0, 0, 0, 0.5]


@pytest.mark.parametrize("grid,mixing,estimator", [
    ((8, 16, 128), "voigt", "epsilon"),       # a z row is one wave
    ((8, 16, 128), "laminate", "epsilon"),    # vector sweeps fused, the direction update a separate sweep
    ((6, 20, 256), "voigt", "residual"),      # a z row is two waves; gamma read back from the device
    ((5, 18, 124), "voigt", "epsilon"),       # tiles with halo lanes, ny not a multiple of the tile height, odd nx
    ((12, 16, 128), "laminate", "residual"),
    ((9, 16, 100), "voigt", "epsilon"),       # rows shorter than a tile (nz / 2 = 50)
    ((6, 14, 84), "laminate", "residual"),
])
def test_fused_cg_matches_oracle_and_unfused(grid, mixing, estimator):
    o = make_oracle(grid, (1.0, 2.0, 1.5), mixing, tol=1e-8, error_estimator=estimator)
    assert o.run_cg(E_LOAD) is False
    res = []
    for fused in (0, 1):
        s = make_gpu_solver(grid, (1.0, 2.0, 1.5), mixing, tol=1e-8, method="cg", error_estimator=estimator, cg_fused=fused)
        assert s.run(E_LOAD) is False
        assert s.iterations == o.iterations
        assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-10
        assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8
        assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-9
        res.append((np.array(s.residuals), s.get_field("epsilon"), s.get_field("u")))
        s.close()
    assert res[0][0].shape == res[1][0].shape and np.abs(res[0][0] - res[1][0]).max() < 1e-12
    assert rel_err(res[1][1], res[0][1]) < 1e-10
    assert np.abs(res[1][2] - res[0][2]).max() < 1e-10 * max(1.0, np.abs(res[0][2]).max())


def test_fused_cg_with_callback_and_maxiter():
    """A convergence callback reads the iterate between iterations (no operator application enqueued ahead); maxiter ends the
    run on the same iterate as the unfused form."""
    grid = (8, 16, 128)
    out = []
    for fused in (0, 1):
        s = make_gpu_solver(grid, mixing="voigt", tol=1e-12, maxiter=7, method="cg", cg_fused=fused)
        seen = []
        s.set_convergence_callback(lambda: seen.append(s.mean_stress().copy()) and False)
        assert s.run(E_LOAD) is False
        out.append((s.iterations, np.array(s.residuals), np.array(seen), s.get_field("epsilon")))
        s.close()
    assert out[0][0] == out[1][0] == 7
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-12
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-11
    assert rel_err(out[1][3], out[0][3]) < 1e-11


@pytest.mark.parametrize("grid,dims,estimator", [((8, 14, 128), (1, 1, 1), "epsilon"), ((6, 16, 256), (2.0, 1.0, 0.5), "epsilon"),
                                                 ((10, 18, 124), (1, 1, 1), "residual"), ((9, 16, 100), (1, 1, 1), "epsilon")])
def test_fused_scalar_cg_matches_oracle_and_unfused(grid, dims, estimator):
    """The same in the scalar modes (CG in potential space): no callback, so the fused form runs."""
    from fibergen_amd import LSSolver
    from oracle.scalar_oracle import ScalarOracle
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 50.0], [1 - phi1, phi1]
    E = np.array([1.0, -0.5, 0.25])
    o = ScalarOracle(*grid, mus=mus, phis=phis, dx=dims[0], dy=dims[1], dz=dims[2], tol=1e-10)
    o.error_estimator = estimator   # "residual": ResidualErrorEstimator F:14382-14405
    assert o.run_cg(E) is False
    res = []
    for fused in (0, 1):
        s = LSSolver(*grid, *dims)
        s.set_options(mode="porous")
        s.set_num_phases(2)
        for p in range(2):
            s.set_phase(p, mus[p], 0.0, phis[p])
        s.set_options(tol=1e-10, method="cg", error_estimator=estimator, cg_fused=fused)
        assert s.run(E) is False
        assert s.iterations == o.iterations
        np.testing.assert_allclose(s.residuals, o.residuals, rtol=0, atol=1e-10)
        assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8
        assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-9
        res.append((np.array(s.residuals), s.get_field("epsilon"), s.get_field("u")))
        s.close()
    assert res[0][0].shape == res[1][0].shape and np.abs(res[0][0] - res[1][0]).max() < 1e-12
    assert rel_err(res[1][1], res[0][1]) < 1e-10
    assert np.abs(res[1][2] - res[0][2]).max() < 1e-10 * max(1.0, np.abs(res[0][2]).max())


def test_scalar_cg_with_callback_equals_run_without():
    """Scalar modes: a convergence callback selects the four-kernel form (the accessors read the iterate in place), a run without
    one the fused sweeps (include/fibergen_amd.h, cg_fused).  The two histories come from different summation orders and must
    agree to the bar that holds between fused and unfused."""
    from fibergen_amd import LSSolver
    grid, dims = (8, 14, 128), (1.0, 2.0, 1.5)
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 50.0], [1 - phi1, phi1]
    E = np.array([1.0, -0.5, 0.25])
    out = []
    for with_cb in (False, True):
        s = LSSolver(*grid, *dims)
        s.set_options(mode="porous")
        s.set_num_phases(2)
        for p in range(2):
            s.set_phase(p, mus[p], 0.0, phis[p])
        s.set_options(tol=1e-10, method="cg", cg_fused=1)
        seen = []
        if with_cb:
            s.set_convergence_callback(lambda: seen.append(s.mean_stress().copy()) and False)
        assert s.run(E) is False
        out.append((s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress().copy(), len(seen)))
        s.close()
    assert out[0][0] == out[1][0] and out[1][4] >= out[1][0] and out[0][4] == 0
    assert out[0][1].shape == out[1][1].shape and np.abs(out[0][1] - out[1][1]).max() < 1e-12
    assert rel_err(out[1][2], out[0][2]) < 1e-10 and rel_err(out[1][3], out[0][3]) < 1e-11


@pytest.mark.parametrize("P", [1, 2, 4])
@pytest.mark.parametrize("mixing", ["voigt", "laminate"])
def test_fused_cg_on_slabs_equals_unfused(P, mixing):
    """The slab driver's CG with the fused sweeps (own planes of the alternate buffers by the tile kernels, their spare planes
    point-wise, so the halo planes stay valid without an exchange) against the four-kernel form and the single-GPU solver."""
    from test_gpu_slab import make_group
    grid = (32, 32, 128)
    ref = make_gpu_solver(grid, mixing=mixing, tol=1e-8, method="cg")
    assert ref.run(E_LOAD) is False
    out = []
    for fused in (0, 1):
        g = make_group(P, grid, mixing=mixing, tol=1e-8, method="cg", cg_fused=fused)
        assert g.run(E_LOAD) is False
        assert g.iterations == ref.iterations
        assert np.abs(np.array(g.residuals) - np.array(ref.residuals)).max() < 1e-11
        assert rel_err(g.get_field("epsilon"), ref.get_field("epsilon")) < 1e-10
        out.append((np.array(g.residuals), g.get_field("epsilon")))
        g.close()
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-12
    assert rel_err(out[1][1], out[0][1]) < 1e-11
    ref.close()


@pytest.mark.parametrize("P,grid", [(1, (8, 16, 128)), (2, (8, 16, 128)), (4, (16, 16, 128))])
def test_fused_scalar_cg_on_slabs_equals_unfused(P, grid):
    """The scalar modes' CG on slabs with the fused sweeps against the host-scalar form."""
    from test_gpu_slab import _scalar_group
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 12.0], [1 - phi1, phi1]
    E = np.array([1.0, -0.5, 0.25])
    out = []
    for fused in (0, 1):
        g = _scalar_group(P, grid, mus, phis, (1.0, 2.0, 1.5), tol=1e-10, method="cg", cg_fused=fused)
        assert g.run(E) is False
        out.append((g.iterations, np.array(g.residuals), g.get_field("epsilon"), g.mean_stress()))
        g.close()
    assert out[0][0] == out[1][0]
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-12
    assert rel_err(out[1][2], out[0][2]) < 1e-10
    assert rel_err(out[1][3], out[0][3]) < 1e-11
