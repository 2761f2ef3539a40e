import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fibergen_amd import LSSolver
from fibergen_amd.rve import bench_rve
from helpers import INCLUSION, MATRIX, lame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mixing = sys.argv[2] if len(sys.argv) > 2 else "voigt"
phi, normals, _ = bench_rve(n, mixing)
for method in ("cg", "basic"):
    s = LSSolver(n, n, n)
    s.set_num_phases(2)
    m0, m1 = lame(**MATRIX), lame(**INCLUSION)
    s.set_phase(0, m0[0], m0[1], 1 - phi); s.set_phase(1, m1[0], m1[1], phi)
    if normals is not None:
        s.set_normals(normals)
    s.set_options(method=method, tol=1e-6, mixing_rule=mixing)
    t0 = time.perf_counter()
    S = np.zeros((6, 6)); its = []
    for i in range(6):
        E = np.zeros(6); E[i] = 1.0
        assert s.run(E) is False
        S[:, i] = s.mean_stress(); its.append(s.iterations)
    dt = time.perf_counter() - t0
    C = S.copy(); C[:, 3:] *= 0.5
    print(method, mixing, n, "six load cases %.2f s, iterations %s, C11 %.6f C12 %.6f C44 %.6f" % (dt, its, C[0, 0], C[0, 1], C[3, 3]))
    s.close()
