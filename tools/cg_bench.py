import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fibergen_amd import LSSolver
from fibergen_amd.rve import bench_rve
from helpers import INCLUSION, MATRIX, lame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
phi, _, _ = bench_rve(n, "voigt")
for method in ("cg", "basic"):
    s = LSSolver(n, n, n)
    s.set_num_phases(2)
    m0, m1 = lame(**MATRIX), lame(**INCLUSION)
    s.set_phase(0, m0[0], m0[1], 1 - phi); s.set_phase(1, m1[0], m1[1], phi)
    s.set_options(method=method, tol=1e-30, abs_tol=0.0, maxiter=30)
    E = np.array([1.0, 0, 0, 0, 0, 0])
    s.run(E)
    t = s.solve_time
    print(method, n, "iterations", s.iterations, "solve_time %.3f s" % t, "it/s %.1f" % (s.iterations / t), "last residual %.3e" % s.residuals[-1])
    s.close()
