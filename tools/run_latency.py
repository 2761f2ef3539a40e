import os, sys, json, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from bench import configure
from fibergen_amd import LSSolver
from fibergen_amd.rve import bench_rve
for n in (64, 128, 256):
    phi, normals, _ = bench_rve(n, "voigt")
    s = LSSolver(n, n, n); configure(s, phi, None, "voigt", "elasticity")
    E = np.array([1.0, 0, 0, 0, 0, 0])
    s.set_options(tol=0.0, abs_tol=0.0, maxiter=200)
    s.run(E); s.run(E)
    it_run = s.iterations / s.solve_time
    s.calc_ref_material(); s.iterate(E, 10); s.synchronize()
    import time; t0 = time.perf_counter(); s.iterate(E, 200); s.synchronize(); it_it = 200 / (time.perf_counter() - t0)
    s.set_options(tol=1e-6, maxiter=10000); s.run(E)
    print(n, "run_load_case %.0f it/s, iterate %.0f it/s; converged run: %d iterations in %.2f ms" % (it_run, it_it, s.iterations, 1e3 * s.solve_time))
    s.close()
