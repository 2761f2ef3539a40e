#!/usr/bin/env python3
"""Run a fibergen XML project on MI355X GPUs -- the drop-in for `fibergen.run()`.

    python examples/run_project.py project.xml                      # one GPU
    torchrun --nproc-per-node 8 examples/run_project.py project.xml # ONE problem, its voxel grid cut into x-slabs over 8 GPUs
    torchrun --nproc-per-node 6 examples/run_project.py project.xml --shard-load-cases
                                                                    # calc_effective_properties: one load case per GPU

Every rank runs the same project and obtains the same results; rank 0 prints them.  Under torchrun the slab decomposition
needs nx and ny divisible by the number of ranks (RCCL all-to-all between the FFT axes over xGMI).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("project")
    ap.add_argument("--shard-load-cases", action="store_true",
                    help="multi-GPU: deal the six load cases of calc_effective_properties to the ranks instead of cutting the grid")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch   # before the HIP library: one shared runtime
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import fibergen   # the reference's module name; fibergen_amd.FG underneath
    fg = fibergen.FG(device=local_rank) if world > 1 else fibergen.FG()
    fg.load_xml(a.project)
    if world > 1:
        if a.shard_load_cases:
            fg.shard_load_cases(True)
        else:
            fg.decompose_slabs(True)
    rc = fg.run()
    if rank == 0:
        print("run() ->", rc, "error" if fg.get_error() else "ok")
        C = fg.get_effective_property()   # [] unless the project ran calc_effective_properties
        if C:
            print("effective property:")
            for row in C:
                print("  " + "  ".join("%12.6g" % v for v in row))
        print("mean stress:", fg.get_mean_stress())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
