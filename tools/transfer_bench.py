"""Boundary transfer rates (fg_set_phase / fg_get_field) with the strided copy (staged_copy=0) and the staged pipeline (1):
    python tools/transfer_bench.py [n=256]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    from fibergen_amd import LSSolver
    rng = np.random.default_rng(0)
    phi = rng.random((n, n, n))
    eps = rng.standard_normal((6, n, n, n))
    for staged in (0, 1, 0, 1):
        s = LSSolver(n, n, n)
        s.set_options(staged_copy=staged)
        s.set_num_phases(2)
        s.set_phase(0, 1.0, 1.0, phi)   # first call: allocations
        s.synchronize()
        out = {"n": n, "staged_copy": staged}
        t0 = time.perf_counter()
        s.set_phase(1, 1.0, 1.0, phi)
        s.synchronize()
        t1 = time.perf_counter()
        out["set_phase_GBps"] = phi.nbytes / (t1 - t0) / 1e9
        t0 = time.perf_counter()
        s.set_field("epsilon", eps)
        s.synchronize()
        t1 = time.perf_counter()
        out["set_field6_GBps"] = eps.nbytes / (t1 - t0) / 1e9
        back = None
        for rep in range(3):
            del back   # (the previous result is unmapped outside the clock)
            t0 = time.perf_counter()
            back = s.get_field("epsilon")
            t1 = time.perf_counter()
            out["get_field6_GBps_%d" % rep] = back.nbytes / (t1 - t0) / 1e9
            out["get_field6_ms_%d" % rep] = 1e3 * (t1 - t0)
        assert np.array_equal(back, eps)
        del back
        s.close()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
