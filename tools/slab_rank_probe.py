"""Kernel times of ONE rank of a P-rank slab decomposition, alone on the GPU: its slab shapes and the blocked all-to-all layouts
of the real run, its data as cache-resident as on a GPU of its own -- the exchanges go to a transport that moves nothing (the
peers' blocks stay what they were: the values are wrong, the kernels and their addresses are the real ones).
    python tools/slab_rank_probe.py [n=256] [P=8] [mixing=voigt] [laminate_overlap=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from bench import configure  # noqa: E402
from fibergen_amd.distributed import SlabMember  # noqa: E402
from fibergen_amd.rve import bench_rve  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mixing = sys.argv[3] if len(sys.argv) > 3 else "voigt"
overlap = int(sys.argv[4]) if len(sys.argv) > 4 else 1
phi, normals, _ = bench_rve(n, mixing, x_range=(0, n // P))
E = np.array([1.0, 0, 0, 0, 0, 0])
for split in (0, 1):
    m = SlabMember(n, n, n, rank=0, nranks=P)
    m.connect_callback(lambda ops: None, lambda values, min_op: None)
    configure(m, phi, normals, mixing, "elasticity")
    m.set_options(slab_split=split, laminate_overlap=overlap)
    m.calc_ref_material()
    m.iterate(E, 5)
    m.synchronize()
    m.enable_stage_timing(True)
    m.iterate(E, 20)
    m.synchronize()
    t, c = m.stage_times()
    m.enable_stage_timing(False)
    st = {k: round(1e3 * v / max(c, 1), 1) for k, v in t.items() if v > 0}
    print("%d^3 %s, rank 0 of %d, slab_split %d, laminate_overlap %d: %s us, sum %.1f us" % (n, mixing, P, split, overlap, st, sum(st.values())), flush=True)
    m.close()
