#!/usr/bin/env python3
"""A grid beyond 2^31 bytes per component, checked without a CPU checker of that size: the periodic 2 x 2 x 2 replica of an
RVE on (2n)^3 voxels has the iterates of the RVE itself on n^3 (same voxel size; the replica's spectrum lives on the even
frequencies), and the n^3 problem is the one the oracle checks (tests/test_gpu_fullsize_oracle.py).  So after k passes the
residual history, the mean stress and every octant of the strain field of the big grid equal the small grid's.
1024^3 takes ~190 GB of the MI355X's 288 GB (strain 6 + polarisation 6 + displacement / force 3 + 3 + phases 2 components
of 8.7 GB).

    python tools/giant_grid_check.py [--n 512] [--passes 4] [--mixing voigt|laminate] > profiles/rNN_giant_grid_<2n>cubed.json
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

E_LOAD = np.array([0.01, -0.003, 0.002, 0.004, -0.001, 0.0025])


def load(a):
    return E_LOAD[:3] if a.mode in ("heat", "porous") else E_LOAD


def solve(n, phi1, normals, cell, a):
    from fibergen_amd import LSSolver
    from helpers import INCLUSION, MATRIX, lame
    s = LSSolver(n, n, n, cell, cell, cell)
    if a.mode != "elasticity":
        s.set_options(mode=a.mode)
    s.set_num_phases(2)
    m0, m1 = (lame(**MATRIX), lame(**INCLUSION)) if a.mode == "elasticity" else ((1.0, 0.0), (10.0, 0.0))
    phi0 = 1.0 - phi1
    s.set_phase(0, m0[0], m0[1], phi0)
    del phi0
    s.set_phase(1, m1[0], m1[1], phi1)
    if normals is not None:
        s.set_normals(normals)
    s.set_options(mixing_rule=a.mixing, method=a.method, tol=-1.0, abs_tol=-1.0, maxiter=a.passes)
    failed = s.run(load(a))
    res = dict(failed=bool(failed), it=int(s.iterations), res=[float(r) for r in s.residuals], ms=s.mean_stress().tolist(),
               me=s.mean_strain().tolist(), ref=[float(v) for v in s.ref_material], vf=float(s.volume_fraction(1)))
    return s, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--mixing", default="voigt", choices=["voigt", "laminate"])
    ap.add_argument("--method", default="basic", choices=["basic", "cg"])
    ap.add_argument("--mode", default="elasticity", choices=["elasticity", "porous", "heat"])
    a = ap.parse_args()
    import psutil
    from fibergen_amd.rve import bench_rve
    n, N = a.n, 2 * a.n
    t0 = time.time()
    phi1, normals, par = bench_rve(n, a.mixing)
    out = {"small": n, "big": N, "mixing": a.mixing, "method": a.method, "mode": a.mode, "passes": a.passes, "rve": par, "rve_s": round(time.time() - t0, 1),
           "host_available_GB": round(psutil.virtual_memory().available / 1e9, 1)}
    s, small = solve(n, phi1, normals, 1.0, a)
    eps_small = s.get_field("epsilon")
    s.close()

    big_phi = np.tile(phi1, (2, 2, 2))
    big_nrm = None if normals is None else np.stack([np.tile(normals[c], (2, 2, 2)) for c in range(3)])
    del phi1, normals
    t0 = time.time()
    s, big = solve(N, big_phi, big_nrm, 2.0, a)   # cell twice as long: the same voxel size
    del big_phi, big_nrm
    out["big_set_up_and_run_s"] = round(time.time() - t0, 1)
    hip = ctypes.CDLL("libamdhip64.so")
    free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b))
    out["device_GB_in_use"] = round((total_b.value - free_b.value) / 1e9, 1)
    out["device_GB_total"] = round(total_b.value / 1e9, 1)

    out["residuals_small"], out["residuals_big"] = small["res"], big["res"]
    out["max_abs_residual_diff"] = float(np.abs(np.array(small["res"]) - np.array(big["res"])).max())
    out["mean_stress_rel_diff"] = float(np.abs(np.array(small["ms"]) - np.array(big["ms"])).max() / np.abs(small["ms"]).max())
    out["mean_strain_abs_diff"] = float(np.abs(np.array(small["me"]) - np.array(big["me"])).max())
    out["ref_material_equal"] = small["ref"] == big["ref"]
    out["volume_fraction_diff"] = abs(small["vf"] - big["vf"])
    ok = (out["max_abs_residual_diff"] < 1e-11 and out["mean_stress_rel_diff"] < 1e-11 and out["mean_strain_abs_diff"] < 1e-13
          and out["ref_material_equal"] and small["failed"] == big["failed"] and small["it"] == big["it"])

    # the strain field, octant by octant (needs 6 N^3 doubles on the host)
    out["field_checked"] = False
    ncomp = 3 if a.mode in ("heat", "porous") else 6
    if psutil.virtual_memory().available > ncomp * 8 * N ** 3 + 16e9:
        eps_big = s.get_field("epsilon")
        scale = float(np.abs(eps_small).max())
        worst = 0.0
        for ox in (0, n):
            for oy in (0, n):
                for oz in (0, n):
                    for c in range(ncomp):
                        d = np.abs(eps_big[c, ox:ox + n, oy:oy + n, oz:oz + n] - eps_small[c]).max()
                        worst = max(worst, float(d))
        del eps_big
        out["field_checked"] = True
        out["strain_octants_max_rel_diff"] = worst / scale
        ok = ok and worst / scale < 1e-11
    del eps_small

    s.synchronize()
    t0 = time.time()
    s.iterate(load(a), 5)
    s.synchronize()
    dt = time.time() - t0
    out["big_ms_per_pass"] = round(dt / 5 * 1e3, 2)
    out["big_it_s"] = round(5 / dt, 3)
    if a.mode == "elasticity":
        out["big_loop_alg_GBps"] = round(296.0 * N ** 3 / (dt / 5) / 1e9, 1)
    s.close()
    out["ok"] = bool(ok)
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
