"""A/B timing of solver options inside ONE job (boxes of the pool differ by up to 15 %):
    python tools/ab.py --n 512 --mixing laminate --set laminate_overlap=0 --set laminate_overlap=1 [--env FG_X=1 ...]
Every --set is one variant (comma-separated key=value pairs); prints it/s (median of 5 x 20 passes) and the kernel table."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--mixing", default="voigt")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--set", action="append", default=[])
    a = ap.parse_args()
    from bench import configure, kernel_table
    from fibergen_amd import LSSolver
    from fibergen_amd.rve import bench_rve
    phi, normals, par = bench_rve(a.n, a.mixing)
    E = np.array([1.0, 0, 0, 0, 0, 0])
    n = (a.n,) * 3
    for variant in (a.set or [""]):
        s = LSSolver(*n)
        configure(s, phi, normals, a.mixing, "elasticity")
        opts = dict(kv.split("=") for kv in variant.split(",") if kv)
        s.set_options(**{k: int(v) for k, v in opts.items()})
        s.calc_ref_material()
        s.iterate(E, 5)
        s.synchronize()
        dts = []
        for _ in range(5):
            t0 = time.perf_counter()
            s.iterate(E, a.steps)
            s.synchronize()
            dts.append(time.perf_counter() - t0)
        ms = [float(v) for v in s.mean_stress()]   # after 5 + 5 * steps passes: equal across variants that change no arithmetic
        kern, _, _ = kernel_table(s, E, n, 10, False)
        med = statistics.median(dts)
        print(json.dumps({"variant": variant, "it_s": a.steps / med, "ms_per_step": 1e3 * med / a.steps,
                          "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in kern.items()},
                          "mean_stress": ms, "env": {k: v for k, v in os.environ.items() if k.startswith("FG_")}}), flush=True)
        s.close()


if __name__ == "__main__":
    main()
