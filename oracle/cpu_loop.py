"""One timed run of the CPU stand-in of the reference's loop (oracle/c_oracle.CRefLoop) in a process of its own, so that the
OpenMP placement variables bench.py sets for it (OMP_NUM_THREADS, OMP_PROC_BIND=spread, OMP_PLACES=cores) are read by a fresh
OpenMP runtime and every buffer is first touched by the threads of THIS thread count.  TEST INFRASTRUCTURE (bench.py's
cpu_baseline leg); prints one JSON line.

    python oracle/cpu_loop.py --grid 256 --mixing voigt --phi phi.npy [--normals n.npy] --threads 64 --max-passes 20 --max-seconds 3
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, required=True)
    ap.add_argument("--mixing", default="voigt")
    ap.add_argument("--phi", required=True)
    ap.add_argument("--normals", default="")
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--max-passes", type=int, default=20)
    ap.add_argument("--max-seconds", type=float, default=3.0)
    ap.add_argument("--fft", default="own", choices=["own", "pocketfft"])
    ap.add_argument("--loops", default="reference", choices=["reference", "contiguous"])
    a = ap.parse_args()
    from helpers import INCLUSION, MATRIX, lame
    from oracle.c_oracle import CRefLoop
    n = (a.grid,) * 3
    phi = np.load(a.phi, mmap_mode="r")
    normals = np.load(a.normals, mmap_mode="r") if a.normals else None
    mats = [lame(**MATRIX), lame(**INCLUSION)]
    E = np.array([1.0, 0, 0, 0, 0, 0])
    mu_0 = 0.5 * (mats[0][0] + mats[1][0])   # any positive reference medium: the cost is identical
    c = CRefLoop(n, (1.0, 1.0, 1.0), mats, [1 - np.asarray(phi), phi], normals, a.mixing, threads=a.threads, fft=a.fft, loops=a.loops)
    c.one_pass(E, mu_0, 0.0)   # warm-up: OpenMP team, page faults of the scratch buffers
    c.fft_seconds = 0.0
    t0 = time.perf_counter()
    it = 0
    while it < a.max_passes and (it < 2 or time.perf_counter() - t0 < a.max_seconds):
        c.one_pass(E, mu_0, 0.0)
        it += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"it_s": it / dt, "fft_share": c.fft_seconds / dt, "passes": it, "threads": a.threads,
                      "fft": "own" if c.own_fft else "pocketfft", "loops": a.loops,
                      "omp": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")}}), flush=True)


if __name__ == "__main__":
    main()
