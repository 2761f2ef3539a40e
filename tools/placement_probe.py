"""Does the time of the displacement sweep depend on WHERE its arrays were allocated?  Several solvers created one after the other
in one process (the allocator hands freed blocks back in changing order), per solver: device addresses of u / f (= the other
buffer) / phi / epsilon / tau and the kernel times of the pass.
    python tools/placement_probe.py [n=512] [solvers=6] [mixing=voigt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from bench import configure  # noqa: E402
from fibergen_amd import LSSolver  # noqa: E402
from fibergen_amd.rve import bench_rve  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mixing = sys.argv[3] if len(sys.argv) > 3 else "voigt"
phi, normals, _ = bench_rve(n, mixing)
E = np.array([1.0, 0, 0, 0, 0, 0])
keep = []
for i in range(count):
    s = LSSolver(n, n, n)
    configure(s, phi, normals, mixing, "elasticity")
    s.calc_ref_material()
    s.iterate(E, 4)
    s.synchronize()
    s.enable_stage_timing(True)
    s.iterate(E, 10)
    s.synchronize()
    t, c = s.stage_times()
    s.enable_stage_timing(False)
    addr = {k: s.device_pointer(k, 0) for k in ("u", "phi", "epsilon", "tau")}
    base = min(addr.values())
    print(i, {k: round(1e3 * v / c, 1) for k, v in t.items() if v > 0},
          {k: "%#x (+%.3f GiB, mod 2MiB %#x)" % (v, (v - base) / 2 ** 30, v % (1 << 21)) for k, v in addr.items()}, flush=True)
    if i % 2 == 0 and os.environ.get("FG_PROBE_KEEP"):
        keep.append(s)      # keep every other solver alive: the next one lands elsewhere
    else:
        s.close()
