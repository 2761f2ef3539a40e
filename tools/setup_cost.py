"""Where the set-up time of FG.run() goes: the Hashin demo project run three times in one process.
    python tools/setup_cost.py [n=64]
Prints per run: wall time of run(), the solver's own solve time (six load cases), and the difference = set-up (project layer,
voxeliser, fg_create, phase upload); then the cost of fg_create + fg_destroy alone for a few grids (first and later calls)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def project(n):
    return """<settings><solver nx="%d" ny="%d" nz="%d"><tol>1e-5</tol><method>cg</method><mixing_rule>voigt</mixing_rule>
      <materials><matrix E="1" nu="0.3" /><coating E="5" nu="0.25" /><inclusion E="10" nu="0.2" /></materials></solver>
      <actions><select_material name="coating" /><place_fiber R="0.4" /><select_material name="inclusion" /><place_fiber R="0.2" />
      <calc_effective_properties /></actions></settings>""" % (n, n, n)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from fibergen_amd import FG, LSSolver
    fg = FG()
    fg.set_xml(project(n))
    for i in range(4):
        t0 = time.perf_counter()
        rc = fg.run()
        t1 = time.perf_counter()
        print("FG.run #%d at %d^3: rc %s, wall %.1f ms, solver %.1f ms, set-up %.1f ms" %
              (i, n, rc, 1e3 * (t1 - t0), 1e3 * fg.get_solve_time(), 1e3 * (t1 - t0 - fg.get_solve_time())), flush=True)
        fg.reset()
        fg.set_xml(project(n))
    for g in (32, 64, 128, 256):
        ts = []
        for i in range(4):
            t0 = time.perf_counter()
            s = LSSolver(g, g, g)
            t1 = time.perf_counter()
            s.close()
            t2 = time.perf_counter()
            ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
        print("%d^3: create / close ms: %s" % (g, ", ".join("%.1f / %.1f" % t for t in ts)), flush=True)
    # pieces of a run at 64^3 on a live solver
    g = 64
    from helpers import INCLUSION, MATRIX, lame, sphere_phi
    phi = sphere_phi((g, g, g), 0.3)
    s = LSSolver(g, g, g)
    s.set_num_phases(2)
    t0 = time.perf_counter()
    s.set_phase(0, *lame(**MATRIX), 1 - phi)
    s.set_phase(1, *lame(**INCLUSION), phi)
    t1 = time.perf_counter()
    s.set_options(tol=1e-5, method="cg")
    for i in range(3):
        t2 = time.perf_counter()
        s.run(np.array([1.0, 0, 0, 0, 0, 0]))
        t3 = time.perf_counter()
        print("64^3 run #%d: wall %.2f ms, solver-reported %.2f ms, %d iterations" % (i, 1e3 * (t3 - t2), 1e3 * s.solve_time, s.iterations))
    print("64^3 set_phase x2: %.2f ms" % (1e3 * (t1 - t0)))
    s.close()


if __name__ == "__main__":
    main()
