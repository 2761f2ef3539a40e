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
