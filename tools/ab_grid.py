"""A/B timing of solver options on a NON-cubic grid inside one job:
    python tools/ab_grid.py --grid 1024,128,128 --set plane_fft=0 --set plane_fft=1 [--mode porous]
Two-phase sphere-in-cell fractions (helpers.sphere_phi), Voigt mixing; prints it/s (median of 5 x steps passes) per variant."""
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
    ap.add_argument("--grid", default="1024,128,128")
    ap.add_argument("--mode", default="elasticity")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--set", action="append", default=[])
    a = ap.parse_args()
    from fibergen_amd import LSSolver
    from helpers import INCLUSION, MATRIX, lame
    grid = tuple(int(v) for v in a.grid.split(","))
    x = [(np.arange(n) + 0.5) / n - 0.5 for n in grid]
    r2 = x[0][:, None, None] ** 2 + x[1][None, :, None] ** 2 + x[2][None, None, :] ** 2
    phi1 = np.clip((0.3 - np.sqrt(r2)) * min(grid) + 0.5, 0.0, 1.0)
    mats = [lame(**MATRIX), lame(**INCLUSION)] if a.mode == "elasticity" else [(1.0, 0.0), (10.0, 0.0)]
    E = np.array([1.0, 0, 0, 0, 0, 0])[: 6 if a.mode != "porous" else 3]
    for variant in (a.set or [""]):
        s = LSSolver(*grid)
        s.set_options(mode=a.mode)
        s.set_num_phases(2)
        s.set_phase(0, mats[0][0], mats[0][1], 1.0 - phi1)
        s.set_phase(1, mats[1][0], mats[1][1], phi1)
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
            dts.append((time.perf_counter() - t0) / a.steps)
        s.enable_stage_timing(True)
        s.iterate(E, a.steps)
        s.synchronize()
        st, cnt = s.stage_times()
        st = {k: v / max(cnt, 1) for k, v in st.items()}
        print(json.dumps({"grid": grid, "mode": a.mode, "variant": variant, "it_s": round(1 / statistics.median(dts), 2),
                          "us_per_pass": round(statistics.median(dts) * 1e6, 1),
                          "stages_us": {k: round(v * 1e3, 1) for k, v in st.items() if isinstance(v, float) and v > 0}}), flush=True)
        s.close()


if __name__ == "__main__":
    main()
