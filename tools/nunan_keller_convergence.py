"""Convergence of the viscosity mode (staggered scheme) towards the Nunan-Keller table the reference carries
(demo/viscosity/nunan_keller/project.xml:21-32) as the grid is refined:  python tools/nunan_keller_convergence.py [nmax=256]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NUNAN_KELLER = {0.01: (0.025941, 0.024813), 0.02: (0.053804, 0.049320), 0.04: (0.11567, 0.097696), 0.08: (0.26755, 0.19337),
                0.12: (0.46580, 0.28995), 0.16: (0.72502, 0.39009), 0.20: (1.0666, 0.49665), 0.24: (1.5228, 0.61306),
                0.28: (2.1459, 0.74379)}


def run(V, n, tol=1e-6, smooth_tol=1e-5):
    from fibergen_amd import FG
    fg = FG()
    fg.set_xml("""
    <settings><print_precision>6</print_precision>
      <solver n="%d">
        <materials><matrix mu="1" /><fiber mu="0" /></materials>
        <mode>viscosity</mode><gamma_scheme>staggered</gamma_scheme><method>cg</method>
        <tol>%g</tol><smooth_tol>%g</smooth_tol><maxiter>20000</maxiter></solver>
      <actions><select_material name="fiber" /><place_fiber V="%g" /><calc_effective_properties /></actions>
    </settings>""" % (n, tol, smooth_tol, V))
    t0 = time.perf_counter()
    assert fg.run() == 0
    mu = fg.get_effective_property()
    a, b = 0.5 * (mu[0][0] - mu[0][1]) - 1, mu[3][3] - 1
    return {"V": V, "n": n, "alpha": a, "beta": b, "alpha_rel": a / NUNAN_KELLER[V][0] - 1, "beta_rel": b / NUNAN_KELLER[V][1] - 1,
            "vf": fg.get_volume_fraction("fiber"), "s": round(time.perf_counter() - t0, 2)}


def main():
    nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for V in (0.08, 0.20):
        n = 16
        while n <= nmax:
            print(json.dumps(run(V, n)), flush=True)
            n *= 2
    for V in sorted(NUNAN_KELLER):
        print(json.dumps(run(V, min(nmax, 128))), flush=True)


if __name__ == "__main__":
    main()
