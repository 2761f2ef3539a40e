"""The slab driver with P members on ONE GPU (in-process group: exchanges are device copies on one stream): what the
decomposition itself costs -- halo planes, blocked y passes, per-slab launches -- without any link in the way.
    python tools/slab_emulation.py --n 256 --members 1,2,4,8"""
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
    ap.add_argument("--members", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from bench import configure
    from fibergen_amd import LSSolver
    from fibergen_amd.distributed import SlabGroup
    from fibergen_amd.rve import bench_rve
    phi, normals, _ = bench_rve(a.n, a.mixing)
    E = np.array([1.0, 0, 0, 0, 0, 0])
    n = (a.n,) * 3

    def timeit(obj):
        obj.calc_ref_material()
        obj.iterate(E, 5)
        obj.synchronize()
        dts = []
        for _ in range(5):
            t0 = time.perf_counter()
            obj.iterate(E, a.steps)
            obj.synchronize()
            dts.append(time.perf_counter() - t0)
        return a.steps / statistics.median(dts)
    s = LSSolver(*n)
    configure(s, phi, normals, a.mixing, "elasticity")
    base = timeit(s)
    s.close()
    out = {"n": a.n, "mixing": a.mixing, "single_gpu_loop_it_s": base, "slab_groups": {}}
    for P in [int(v) for v in a.members.split(",")]:
        for split in ((0, 1) if P > 1 else (0,)):
            g = SlabGroup(*n, nranks=P)
            configure(g, phi, normals, a.mixing, "elasticity")
            g.set_options(slab_split=split)
            v = timeit(g)
            out["slab_groups"]["P=%d split=%d" % (P, split)] = {"it_s": v, "ratio": v / base}
            g.close()
    # the RCCL transport in loop-back: one slab whose blocks / planes / norms go through ncclSend / ncclRecv / ncclAllReduce to
    # itself on the second stream -- what the RCCL calls themselves cost per pass (on-device copies, no link)
    from fibergen_amd.distributed import SlabMember, rccl_unique_id
    out["rccl_loopback"] = {}
    for split in (0, 1):
        m = SlabMember(*n, rank=0, nranks=1)
        m.connect_rccl(rccl_unique_id())
        configure(m, phi, normals, a.mixing, "elasticity")
        m.set_options(slab_loopback=1, slab_split=split)
        v = timeit(m)
        out["rccl_loopback"]["split=%d" % split] = {"it_s": v, "ratio": v / base}
        m.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
