"""Six load cases of calc_effective_properties on ONE GPU: one after the other against six solvers on six streams driven by six
threads (small grids underfill the GPU).  python tools/concurrent_cases_probe.py [n]"""
import sys, os, time, threading
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from fibergen_amd import LSSolver
from helpers import sphere_phi, lame, MATRIX, INCLUSION

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
method = sys.argv[2] if len(sys.argv) > 2 else "cg"
phi = sphere_phi((n, n, n), 0.3)
m0, m1 = lame(**MATRIX), lame(**INCLUSION)


def make():
    s = LSSolver(n, n, n)
    s.set_num_phases(2)
    s.set_phase(0, m0[0], m0[1], 1 - phi)
    s.set_phase(1, m1[0], m1[1], phi)
    s.set_options(tol=1e-6, method=method)
    return s


loads = [np.eye(6)[i] for i in range(6)]
s = make()
s.run(loads[0])          # warm-up
t0 = time.time()
its = []
for E in loads:
    s.run(E)
    its.append(s.iterations)
t_seq = time.time() - t0
s.close()
ss = [make() for _ in range(6)]
for x in ss:
    x.run(loads[0])
t0 = time.time()
th = [threading.Thread(target=ss[i].run, args=(loads[i],)) for i in range(6)]
for t in th:
    t.start()
for t in th:
    t.join()
t_par = time.time() - t0
print("n=%d method=%s iterations %s: sequential %.1f ms, six threads / streams %.1f ms, x%.2f" % (n, method, its, 1e3 * t_seq, 1e3 * t_par, t_seq / t_par))
