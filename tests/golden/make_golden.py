#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the NumPy oracle (oracle/ls_oracle.py).

The reference holds no stored numeric fixtures and cannot be built here, so
these vectors freeze the *pinned oracle* (tests/test_oracle_pins.py) on seeded inputs:
per-stage inputs/outputs on 8x6x4 and 12x10x7 grids and full-run scalars on 16^3 sphere RVEs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import make_oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def stages(grid, dims, seed):
    rng = np.random.default_rng(seed)
    d = {"grid": np.array(grid), "dims": np.array(dims, dtype=np.float64)}
    nzc = grid[2] // 2 + 1
    d["eps"] = rng.standard_normal((6,) + grid)
    d["tau"] = rng.standard_normal((6,) + grid)
    d["u"] = rng.standard_normal((3,) + grid)
    d["E"] = np.array([0.3, -0.2, 0.1, 0.05, -0.07, 0.02])
    d["spec"] = rng.standard_normal((3,) + grid[:2] + (nzc,)) + 1j * rng.standard_normal((3,) + grid[:2] + (nzc,))
    d["mu_0"], d["lambda_0"] = np.float64(0.77), np.float64(0.31)
    for mixing in ("voigt", "laminate"):
        o = make_oracle(grid, dims, mixing)
        d["stress_" + mixing] = o.calc_stress(0.77, 0.31, d["eps"])
        d["sigma_" + mixing] = o.calc_stress(0.0, 0.0, d["eps"])
        o.calc_ref_material()
        d["ref_mu_0_" + mixing] = np.float64(o.mu_0)
        d["iter_" + mixing] = o.basic_scheme(d["E"], 0.1 * d["eps"])
    o = make_oracle(grid, dims)
    d["div"] = o.div_staggered(d["tau"])
    d["epsop"] = o.eps_staggered(d["E"], d["u"])
    d["fft"] = o.fft_vector(d["u"])
    d["ifft"] = o.ifft_vector(d["spec"])
    d["g0_m1"] = o.g0_apply(1324.3, 324.2, d["spec"], -1.0)
    d["g0_p1"] = o.g0_apply(1324.3, 324.2, d["spec"], 1.0)
    return d


def full_run(grid, mixing, tol=1e-8):
    o = make_oracle(grid, mixing=mixing, tol=tol)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    assert o.run(E) is False
    d = {"grid": np.array(grid), "E": E, "tol": np.float64(tol), "iterations": np.int64(o.iterations),
         "residuals": np.array(o.residuals), "mean_stress": o.mean_stress(), "mean_strain": o.mean_strain(),
         "mu_0": np.float64(o.mu_0), "eps_x0": o.eps[:, 0].copy(), "sigma_x0": o.get_field("sigma")[:, 0].copy()}
    d["Ceff_voigt"] = o.calc_effective_properties()
    d["ceff_iterations"] = np.array(o.ceff_iterations)
    return d


if __name__ == "__main__":
    np.savez_compressed(os.path.join(OUT, "stages_8x6x4.npz"), **stages((8, 6, 4), (1.0, 2.0, 3.0), 100))
    np.savez_compressed(os.path.join(OUT, "stages_12x10x7.npz"), **stages((12, 10, 7), (1.0, 1.0, 1.0), 101))
    np.savez_compressed(os.path.join(OUT, "stages_16x8x16.npz"), **stages((16, 8, 16), (1.0, 1.0, 1.0), 102))
    for mixing in ("voigt", "laminate"):
        np.savez_compressed(os.path.join(OUT, "run_16cubed_%s.npz" % mixing), **full_run((16, 16, 16), mixing))
    print("golden fixtures written to", OUT)
