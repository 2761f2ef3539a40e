"""GPU parity of the scalar modes (mode=heat / porous, BASELINE config 5): the potential-based HIP loop through the
C ABI against oracle/scalar_oracle.py on the same phase fields."""
import numpy as np
import pytest

from helpers import rel_err, sphere_phi

pytestmark = pytest.mark.gpu


def _solver(grid, mus, phis, dims=(1.0, 1.0, 1.0), **kw):
    from fibergen_amd import LSSolver
    s = LSSolver(*grid, *dims)
    s.set_options(mode="porous")
    s.set_num_phases(len(mus))
    for p, (mu, phi) in enumerate(zip(mus, phis)):
        s.set_phase(p, mu, 0.0, phi)
    s.set_options(**kw)
    return s


def _oracle(grid, mus, phis, dims=(1.0, 1.0, 1.0), **kw):
    from oracle.scalar_oracle import ScalarOracle
    return ScalarOracle(*grid, mus=mus, phis=phis, dx=dims[0], dy=dims[1], dz=dims[2], **kw)


@pytest.mark.parametrize("grid,dims", [((16, 16, 16), (1, 1, 1)), ((12, 10, 6), (2.0, 1.0, 0.5)), ((9, 7, 5), (1, 1, 1)),
                                       ((32, 16, 64), (1, 1, 1)), ((8, 16, 5), (1, 1, 1))])
@pytest.mark.parametrize("u_loop", [1, 2])
def test_scalar_run_matches_oracle(grid, dims, u_loop):
    """u_loop = 1: per-phase accumulation in the reference's order; 2 (default): precomputed effective
    conductivity + FMA + lane shifts."""
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 12.0], [1 - phi1, phi1]
    E = np.array([1.0, -0.5, 0.25])
    s = _solver(grid, mus, phis, dims, tol=1e-9, u_loop=u_loop)
    o = _oracle(grid, mus, phis, dims, tol=1e-9)
    assert s.run(E) is False and o.run(E) is False
    assert s.iterations == o.iterations
    assert s.ref_material[0] == o.mu_0
    np.testing.assert_allclose(s.residuals, o.residuals, rtol=0, atol=1e-11)
    g = s.get_field("epsilon")
    assert g.shape == (3,) + grid
    assert rel_err(g, o.eps) < 1e-10
    assert rel_err(s.get_field("sigma"), o.pk1(o.eps)) < 1e-10
    assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-11
    np.testing.assert_allclose(s.mean_strain(), E, atol=1e-12)
    T = s.get_field("u")
    assert T.shape == (1,) + grid
    To = o.potential()
    assert np.abs(T[0] - To).max() < 1e-10 * max(1.0, np.abs(To).max())
    # a second load case on the same object, then raw passes continue from the state
    assert s.run([0, 0, 1.0]) is False and o.run([0, 0, 1.0]) is False
    assert s.iterations == o.iterations and rel_err(s.get_field("epsilon"), o.eps) < 1e-10
    s.iterate([0, 0, 1.0], 2)
    g2 = o.basic_scheme(np.array([0, 0, 1.0]), o.basic_scheme(np.array([0, 0, 1.0]), o.eps))
    assert rel_err(s.get_field("epsilon"), g2) < 1e-10
    s.close()


@pytest.mark.parametrize("grid,dims", [((16, 16, 128), (1, 1, 1)), ((8, 14, 124), (1.0, 2.0, 0.5)), ((6, 20, 130), (1, 1, 1)),
                                       ((40, 30, 128), (1, 1, 1)), ((12, 16, 256), (1, 1, 1)), ((5, 14, 256), (2, 1, 1)),
                                       ((16, 16, 100), (1, 1, 1)), ((9, 14, 80), (1.0, 2.0, 0.5)), ((7, 20, 122), (1, 1, 1))])
def test_tiled_scalar_sweep(grid, dims):
    """u_loop = 2 on grids the LDS-tiled marching sweep k_sc_tile takes (nz/2 >= 40: halo lanes; nz/2 = 64 / 128: whole
    rows of one / two waves; overlapping last tiles, periodic wrap in all directions): same iterates as the exact-order
    sweep (u_loop = 1) and as the oracle."""
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 12.0], [1 - phi1, phi1]
    E = np.array([1.0, -0.5, 0.25])
    res = {}
    for flag in (1, 2):
        s = _solver(grid, mus, phis, dims, tol=1e-9, u_loop=flag)
        assert s.run(E) is False
        res[flag] = (s.iterations, np.array(s.residuals), s.get_field("epsilon"), s.mean_stress(), s.get_field("u"))
        s.close()
    a, b = res[1], res[2]
    assert a[0] == b[0]
    assert np.abs(a[1] - b[1]).max() < 1e-12
    assert rel_err(b[2], a[2]) < 1e-11 and rel_err(b[3], a[3]) < 1e-12
    assert np.abs(b[4] - a[4]).max() < 1e-11 * max(1.0, np.abs(a[4]).max())
    o = _oracle(grid, mus, phis, dims, tol=1e-9)
    assert o.run(E) is False
    assert o.iterations == b[0] and rel_err(b[2], o.eps) < 1e-10


def test_scalar_effective_conductivity_of_layers_is_exact():
    """series / parallel means (see tests/test_oracle_pins.py) from the HIP path"""
    shape, mus, fr = (20, 4, 6), [1.0, 5.0, 0.5], [0.2, 0.3, 0.5]
    edges = np.round(np.cumsum([0.0] + fr) * shape[0]).astype(int)
    phis = []
    for a, b in zip(edges[:-1], edges[1:]):
        p = np.zeros(shape)
        p[a:b] = 1.0
        phis.append(p)
    s = _solver(shape, mus, phis, tol=1e-13, maxiter=2000)
    K = np.zeros((3, 3))
    for i in range(3):
        assert s.run(np.eye(3)[i]) is False
        K[:, i] = s.mean_stress()
    assert K[0, 0] == pytest.approx(1.0 / sum(f / m for f, m in zip(fr, mus)), rel=1e-10)
    assert K[1, 1] == pytest.approx(sum(f * m for f, m in zip(fr, mus)), rel=1e-12)
    assert np.abs(K - np.diag(np.diag(K))).max() < 1e-12


def test_scalar_callback_and_errors():
    grid = (16, 16, 16)
    phi1 = sphere_phi(grid, 0.3)
    s = _solver(grid, [1.0, 12.0], [1 - phi1, phi1], tol=1e-9)
    o = _oracle(grid, [1.0, 12.0], [1 - phi1, phi1], tol=1e-9)
    seen = []

    def cb():
        seen.append((s.get_field("u").copy(), s.mean_stress().copy()))
        return False
    s.set_convergence_callback(cb)
    assert s.run([1.0, 0, 0]) is False and o.run([1.0, 0, 0]) is False
    assert s.iterations == o.iterations == len(seen)
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-10
    assert rel_err(seen[-1][1], o.mean_stress()) < 1e-10
    s.set_convergence_callback(None)
    s.set_options(mixing_rule="laminate")
    with pytest.raises(RuntimeError, match="Voigt"):
        s.run([1.0, 0, 0])
    with pytest.raises(RuntimeError):
        s.set_field("epsilon", np.zeros((3,) + grid))
    s.close()


POROUS_XML = """
<settings><solver n="16"><mode>%s</mode><tol>1e-9</tol>
  <materials><matrix mu="1" /><incl mu="12" /></materials></solver>
  <actions><select_material name="incl" /><place_fiber R="0.3" />
  <run_load_case e1="1" e3="0.5" outfile="%s" />
  <calc_effective_properties /></actions></settings>"""


@pytest.mark.parametrize("mode", ["heat", "porous"])
def test_fg_scalar_project(tmp_path, mode):
    """XML project in heat / porous mode through FG: 3x3 effective matrix = the oracle's on the same phase field;
    result file with gradient, flux and potential (writeVTK heat branch  F:23436-23450)."""
    from fibergen_amd import FG, vtk
    fn = str(tmp_path / "r.vtk")
    fg = FG()
    fg.set_xml(POROUS_XML % (mode, fn))
    assert fg.run() == 0
    K = np.array(fg.get_effective_property())
    assert K.shape == (3, 3)
    phi = fg.get_field("phi")
    o = _oracle((16, 16, 16), [1.0, 12.0], [phi[0], phi[1]], tol=1e-9)
    Ko = np.zeros((3, 3))
    for i in range(3):   # the project does not name a method: the reference's default, cg
        assert o.run_cg(np.eye(3)[i]) is False
        Ko[:, i] = o.mean_stress()
    assert rel_err(K, Ko) < 1e-9
    assert fg._lss.iterations == o.iterations
    assert fg.get_field("epsilon").shape == (3, 16, 16, 16) and fg.get_field("u").shape == (1, 16, 16, 16)
    assert len(fg.get_mean_stress()) == 3
    h, f = vtk.read_legacy(fn)
    assert list(f) == ["phi_matrix", "phi_incl", "epsilon_11", "epsilon_22", "epsilon_33", "sigma_11", "sigma_22",
                       "sigma_33", "T" if mode == "heat" else "p"]
    assert abs(f["epsilon_11"].mean() - 1.0) < 1e-6 and abs(f["epsilon_33"].mean() - 0.5) < 1e-6


@pytest.mark.parametrize("grid,dims", [((16, 16, 16), (1, 1, 1)), ((12, 10, 6), (2.0, 1.0, 0.5)), ((8, 14, 128), (1, 1, 1))])
def test_scalar_cg_matches_oracle(grid, dims):
    """method=cg in the scalar modes (runCG -> runCGElasticity with 3 components, F:22056-22066) carried in potential space:
    iteration count, residual history, fields and mean flux of the oracle's restatement; far fewer iterations than the
    basic scheme; accessors in the callback see the iterate."""
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 50.0], [1 - phi1, phi1]
    E = np.array([1.0, -0.5, 0.25])
    s = _solver(grid, mus, phis, dims, tol=1e-10, method="cg")
    o = _oracle(grid, mus, phis, dims, tol=1e-10)
    seen = []

    def cb():
        seen.append((s.get_field("u").copy(), s.mean_stress().copy()))
        return False
    s.set_convergence_callback(cb)
    assert s.run(E) is False and o.run_cg(E) is False
    assert s.iterations == o.iterations and len(seen) == len(o.residuals)
    np.testing.assert_allclose(s.residuals, o.residuals, rtol=0, atol=1e-10)
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8
    assert rel_err(s.mean_stress(), o.mean_stress()) < 1e-9
    assert rel_err(seen[-1][1], o.mean_stress()) < 1e-9
    b = _solver(grid, mus, phis, dims, tol=1e-10)
    assert b.run(E) is False
    assert s.iterations < b.iterations / 3
    assert rel_err(s.mean_stress(), b.mean_stress()) < 1e-6
    s.close()
    b.close()


@pytest.mark.parametrize("grid,u_loop", [((16, 16, 16), 1), ((12, 10, 6), 2),
                                         # grids the LDS-tiled sweep takes: it carries the sums of the flux polarisation itself
                                         ((8, 16, 128), 2), ((6, 14, 256), 2), ((8, 14, 124), 2)])
def test_scalar_mixed_boundary_conditions(grid, u_loop):
    """initBCProjector / applyBCProjector of GammaOperatorStaggeredHeat  F:20342-20350 with setBCProjector for dim 3
    (F:20599-20665): a flux prescribed in x, gradients prescribed in y and z."""
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 12.0], [1 - phi1, phi1]
    P3 = np.diag([0.0, 1.0, 1.0])
    P6 = np.diag([0.0, 1.0, 1.0, 0.5, 0.5, 0.5])
    E, S = np.array([0.0, 0.3, -0.2]), np.array([1.5, 0.0, 0.0])
    s = _solver(grid, mus, phis, tol=1e-10, bc_tol=1e-9, maxiter=500, u_loop=u_loop)
    s.set_bc_projector(P6)
    o = _oracle(grid, mus, phis, tol=1e-10, bc_tol=1e-9, maxiter=500)
    assert o.run(E, S, P3) is False and s.run(E, S) is False
    assert s.iterations == o.iterations
    assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-10
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-9
    assert s.mean_stress()[0] == pytest.approx(1.5, rel=1e-8) and np.abs(s.mean_strain()[1:] - E[1:]).max() < 1e-12
    s.set_options(method="cg")
    with pytest.raises(RuntimeError, match="method=basic"):
        s.run(E, S)
    s.close()


@pytest.mark.parametrize("grid", [(20, 12, 14), (100, 6, 18), (120, 4, 10), (200, 3, 12), (300, 3, 8), (225, 2, 6), (500, 2, 4)])
def test_scalar_modes_on_decimal_grids_fused_x_pass(grid):
    """the one-component fused x pass of the tile kernels (k_smooth_xjoint<..., 1>: kernels per largest radix; joint_x = 0: the
    first form, k_smooth_xfused) against the scalar oracle"""
    phi1 = sphere_phi(grid, 0.3)
    mus, phis = [1.0, 12.0], [1 - phi1, phi1]
    E = np.array([1.0, -0.5, 0.25])
    o = _oracle(grid, mus, phis, tol=1e-9)
    assert o.run(E) is False
    for joint in (1, 0):
        s = _solver(grid, mus, phis, tol=1e-9, joint_x=joint)
        assert s.run(E) is False
        assert s.iterations == o.iterations
        np.testing.assert_allclose(s.residuals, o.residuals, rtol=0, atol=1e-11)
        assert rel_err(s.get_field("epsilon"), o.eps) < 1e-10
        s.close()


def test_scalar_fused_x_plan_kernels_equal_the_class_kernels():
    """every length in the one-component table of fg_fft_smooth_plans.h as the x axis of a thin grid: three passes of the porous
    loop with the plan kernels (default) and with the class kernels (tile_plans = 0) give the same gradient field"""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fibergen_amd", "csrc",
                            "fg_fft_smooth_plans.h")).read()
    body = src[src.index("#define FG_SMOOTH_X1_PLANS(X)"):]
    lengths = sorted({int(m.group(1)) for m in re.finditer(r"X\((\d+),", body[:body.index("\n\n")])})
    assert len(lengths) >= 38
    E = np.array([1.0, -0.5, 0.25])
    try:
        for n in lengths:
            grid = (n, 3, 10)
            phi1 = sphere_phi(grid, 0.3)
            out = {}
            for flag in (1, 0):
                s = _solver(grid, [1.0, 12.0], [1 - phi1, phi1], tile_plans=flag)
                s.calc_ref_material()
                s.iterate(E, 3)
                out[flag] = s.get_field("epsilon")
                s.close()
            assert rel_err(out[1], out[0]) < 1e-12, grid
    finally:
        s = _solver((8, 8, 8), [1.0, 12.0], [0.5 * np.ones((8, 8, 8))] * 2, tile_plans=1)
        s.close()
