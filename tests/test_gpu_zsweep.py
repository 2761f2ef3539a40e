"""The displacement sweep with both z transforms attached (fg_kernels_zsweep.hip, option z_sweep): the state between passes is
the z half spectrum of u, so neither u nor f exists in real space in memory.  Same arithmetic as the tiled sweep + the c2r /
r2c passes it absorbs (the line transforms use the four-point radix-4 schedule instead of radix 8): results equal the
separate-pass loop to FFT rounding, iteration counts and residual histories are the oracle's."""
import numpy as np
import pytest

from helpers import make_gpu_solver, make_oracle, rel_err

pytestmark = pytest.mark.gpu

E_LOAD = [1.0, 0, 0, 0, 0, 0.5]


def run_pair(grid, dims=(1.0, 1.0, 1.0), **opts):
    out = []
    for z in (0, 1):
        s = make_gpu_solver(grid, dims, "voigt", z_sweep=z, **opts)
        failed = s.run(E_LOAD)
        assert (s.counter("zsweep_passes") > 0) == bool(z)   # the sweep under test really ran (and only where asked for)
        out.append((failed, s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress(), s.get_field("u")))
        s.close()
    return out


@pytest.mark.parametrize("grid,dims", [
    ((8, 16, 128), (1.0, 1.0, 1.0)),      # nz/2 = 64: one wave per z row, ONE y tile, marches shorter than the grid
    ((16, 24, 128), (1.0, 2.0, 1.5)),     # ny not a multiple of the tile height (overlapping last tile), anisotropic cell
    ((12, 16, 256), (1.0, 1.0, 1.0)),     # nz/2 = 128: two waves per z row
    ((20, 44, 256), (2.0, 1.0, 0.5)),
    ((4, 16, 256), (1.0, 1.0, 1.0)),      # the shortest march
])
def test_zsweep_run_matches_separate_passes(grid, dims):
    (f0, it0, r0, e0, s0, u0), (f1, it1, r1, e1, s1, u1) = run_pair(grid, dims, tol=1e-9)
    assert f0 is False and f1 is False
    assert it0 == it1
    assert np.abs(r0 - r1).max() < 1e-11
    assert rel_err(e1, e0) < 1e-11
    assert rel_err(s1, s0) < 1e-12
    assert np.abs(u1 - u0).max() < 1e-11 * max(1.0, np.abs(u0).max())


@pytest.mark.parametrize("grid", [(8, 16, 128), (12, 20, 256)])
def test_zsweep_matches_oracle(grid):
    o = make_oracle(grid, (1.0, 1.0, 1.0), "voigt", tol=1e-8)
    assert o.run(E_LOAD) is False
    s = make_gpu_solver(grid, (1.0, 1.0, 1.0), "voigt", z_sweep=1, tol=1e-8)
    assert s.run(E_LOAD) is False
    assert s.counter("zsweep_passes") >= s.iterations - 1
    assert s.iterations == o.iterations
    assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-11
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-10
    assert rel_err(s.get_field("sigma"), o.get_field("sigma")) < 1e-10
    assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-11
    s.close()


def test_zsweep_iterate_and_accessors_between_passes():
    """fg_iterate leaves the spectrum state; accessors bring u back to real space and the loop re-enters from there."""
    grid = (8, 16, 256)
    a = make_gpu_solver(grid, mixing="voigt", z_sweep=0)
    b = make_gpu_solver(grid, mixing="voigt", z_sweep=1)
    for s in (a, b):
        s.calc_ref_material()
        s.iterate(E_LOAD, 3)
    assert rel_err(b.get_field("epsilon"), a.get_field("epsilon")) < 1e-12
    for s in (a, b):
        s.iterate(E_LOAD, 2)                      # re-enters the spectrum loop from the real-space state
    assert rel_err(b.mean_stress(), a.mean_stress()) < 1e-12
    for s in (a, b):
        s.iterate(E_LOAD, 4)
    assert a.counter("zsweep_passes") == 0 and b.counter("zsweep_passes") == 8
    assert rel_err(b.get_field("u"), a.get_field("u")) < 1e-11
    assert rel_err(b.get_field("epsilon"), a.get_field("epsilon")) < 1e-12
    a.close()
    b.close()


def test_zsweep_mixed_bc():
    """Mixed boundary conditions: the sums of the polarisation come out of the same sweep (SUMT)."""
    grid = (8, 16, 128)
    res = []
    for z in (0, 1):
        s = make_gpu_solver(grid, mixing="voigt", z_sweep=z, tol=1e-9, bc_tol=1e-8, maxiter=400)
        P = np.zeros((6, 6))
        P[0, 0] = 1.0
        s.set_bc_projector(P)
        assert s.run([0.01, 0, 0, 0, 0, 0], np.zeros(6)) is False
        assert (s.counter("zsweep_passes") > 0) == bool(z)
        res.append((s.iterations, s.get_field("epsilon"), s.mean_stress()))
        s.close()
    assert res[0][0] == res[1][0]
    assert rel_err(res[1][1], res[0][1]) < 1e-10
    assert np.abs(res[1][2] - res[0][2]).max() < 1e-12


def test_zsweep_three_phases():
    """Three phases (no complementary pair): the sweep reads the two effective-moduli arrays."""
    from fibergen_amd import LSSolver
    from helpers import sphere_phi, lame
    grid = (8, 16, 128)
    p1 = sphere_phi(grid, 0.25, (0.3, 0.5, 0.5))
    p2 = sphere_phi(grid, 0.2, (0.75, 0.5, 0.5))
    res = []
    for z in (0, 1):
        s = LSSolver(*grid, 1.0, 1.0, 1.0)
        s.set_num_phases(3)
        s.set_phase(0, *lame(1.0, 0.3), 1.0 - p1 - p2)
        s.set_phase(1, *lame(10.0, 0.2), p1)
        s.set_phase(2, *lame(4.0, 0.25), p2)
        s.set_options(z_sweep=z, tol=1e-9)
        assert s.run(E_LOAD) is False
        assert (s.counter("zsweep_passes") > 0) == bool(z)
        res.append((s.iterations, s.get_field("epsilon")))
        s.close()
    assert res[0][0] == res[1][0]
    assert rel_err(res[1][1], res[0][1]) < 1e-11
