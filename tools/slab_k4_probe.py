"""Kernel times of ONE slab of an 8-slab group at 256^3 (what each of 8 GPUs runs per pass), all slabs in this process:
    FG_XF_SLAB=0/1 python tools/slab_k4_probe.py [n=256] [P=8]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from bench import configure  # noqa: E402
from fibergen_amd.distributed import SlabGroup  # noqa: E402
from fibergen_amd.rve import bench_rve  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
phi, normals, _ = bench_rve(n, "voigt")
E = np.array([1.0, 0, 0, 0, 0, 0])
for split in (0, 1):
    g = SlabGroup(n, n, n, nranks=P)
    configure(g, phi, normals, "voigt", "elasticity")
    g.set_options(slab_split=split)
    g.calc_ref_material()
    g.iterate(E, 5)
    g.synchronize()
    m = g.members[0]
    m.enable_stage_timing(True)
    g.iterate(E, 20)
    g.synchronize()
    t, c = m.stage_times()
    m.enable_stage_timing(False)
    print(os.environ.get("FG_XF_SLAB"), "split", split, {k: round(1e3 * v / max(c, 1), 1) for k, v in t.items() if v > 0}, "us", flush=True)
    g.close()
