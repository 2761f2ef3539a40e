"""Host-side cost of enqueueing one pass of the slab driver with the RCCL transport and P ranks (all on ONE GPU, each
posing as its own host: see tests/test_gpu_distributed.py).  The bytes cross loop-back sockets, so the GPU-side time means
nothing; what is measured is how long the host needs to ISSUE a pass (kernel launches, event calls, RCCL group calls with
2 x 3 (P - 1) sends and receives per all-to-all) -- the floor of the pass time on a real node.
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/slab_enqueue_cost.py [n=256]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank = int(os.environ.get("RANK", "0"))
os.environ["NCCL_HOSTID"] = "fibergen-enqueue-rank-%d" % rank
os.environ["NCCL_SOCKET_IFNAME"] = "lo"
os.environ["NCCL_IB_DISABLE"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

dist.init_process_group("gloo")
from bench import configure  # noqa: E402
from fibergen_amd.distributed import DistributedLSSolver  # noqa: E402
from fibergen_amd.rve import bench_rve  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mixing = sys.argv[2] if len(sys.argv) > 2 else "voigt"
phi, normals, _ = bench_rve(n, mixing)
d = DistributedLSSolver(n, n, n, device=0, transport="rccl")
configure(d, phi, normals, mixing, "elasticity", slab=d.slab)
d.calc_ref_material()
E = np.array([1.0, 0, 0, 0, 0, 0])
d.iterate(E, 3)
d.synchronize()
dist.barrier()
out = []
for steps in (5, 10):
    t0 = time.perf_counter()
    d.iterate(E, steps)          # returns when everything is enqueued
    t1 = time.perf_counter()
    d.synchronize()
    t2 = time.perf_counter()
    out.append((steps, 1e6 * (t1 - t0) / steps, 1e6 * (t2 - t0) / steps))
    dist.barrier()
if rank == 0:
    for steps, enq, tot in out:
        print("P=%d n=%d %s: host enqueue %.0f us per pass (%d passes); with the socket transfers %.0f us per pass" %
              (dist.get_world_size(), n, mixing, enq, steps, tot), flush=True)
d.close()
dist.barrier()
dist.destroy_process_group()
