#!/usr/bin/env python3
"""Benchmark of the Lippmann-Schwinger basic-scheme pass on MI355X.

    python bench.py --gpus N --steps K --warmup W [--n 256] [--mixing voigt|laminate]

A "step" is one LS iteration (one basicScheme pass, F:20558-20578) on a synthetic fibre
RVE resident in HBM.  With N > 1 (torchrun, one rank per GPU) every rank solves its own
load case of the same RVE -- the six load cases of calc_effective_properties
(F:26030-26114) are independent, so this shards without any data-path collective
("scaling": "weak"); the value is the aggregate over ranks.
Rank 0 prints one JSON line (metric of BASELINE.json: LS iterations/s, plus the HBM
roofline of the dominant kernel and the CPU baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(n, nphases, mixing):
    """Algorithmic HBM bytes per launch of each kernel (DESIGN.md / SURVEY 8d): inputs read
    once, outputs written once, float64."""
    nx, ny, nz = n
    N = nx * ny * nz
    nzc = nz // 2 + 1
    F = nx * ny * nzc  # complex frequencies (= padded pairs)
    stress = (6 + 1 + 6) * 8 * N  # 6 eps + phi (second phase = 1 - phi) in, 6 tau out
    if nphases != 2:
        stress = (12 + nphases) * 8 * N
    fft_pass = 3 * 2 * 16 * F     # 3 components, read + write, complex128
    return {
        "stress": stress,
        "div": 9 * 8 * N,
        "r2c_z": fft_pass, "c2c_y_fwd": fft_pass, "c2c_x_fwd": fft_pass,
        "g0": 96 * F,
        "c2c_x_inv": fft_pass, "c2c_y_inv": fft_pass, "c2r_z": fft_pass,
        "eps_norm": 9 * 8 * N,
    }


A_MIN_BYTES_PER_VOXEL = 392   # maximum legal fusion
A_STAGE_BYTES_PER_VOXEL = 632


def cpu_baseline(n, mixing, budget_s=20.0):
    """The reference's per-iteration loop nests restated in C/OpenMP (oracle/c, pass structure of
    BASELINE.md section 3) + pocketfft (scipy.fft, workers = cores) in place of threaded FFTW,
    timed on all host cores on a bounded sample: passes of the same RVE until ~budget_s is used."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.c_oracle import CRef
    from fibergen_amd.rve import synthetic_fiber_rve
    from helpers import INCLUSION, MATRIX, lame
    scale = max(n[0], 128) / 128.0
    K = int(round(40 * scale ** 3)) if n[0] >= 128 else 5
    phi, normals = synthetic_fiber_rve(n, K=K, R=0.05 / scale, L=0.4 / scale, seed=0,
                                       with_normals=(mixing == "laminate"))
    c = CRef(n, (1.0, 1.0, 1.0), [lame(**MATRIX), lame(**INCLUSION)], [1 - phi, phi], normals, mixing)
    E = np.array([1.0, 0, 0, 0, 0, 0])
    eps = np.zeros((6,) + tuple(n))
    mu_0 = 0.5 * (lame(**MATRIX)[0] + lame(**INCLUSION)[0])  # any positive reference medium: cost is identical
    eps = c.basic_scheme(E, eps, mu_0, 0.0)   # warm-up (page faults, pocketfft plan cache)
    c.fft_seconds = 0.0
    t0 = time.perf_counter()
    it = 0
    while True:
        eps = c.basic_scheme(E, eps, mu_0, 0.0)
        c.component_norm(eps)
        it += 1
        if time.perf_counter() - t0 > budget_s or it >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": it / dt, "unit": "it/s", "cores": int(c.threads), "kind": "port",
            "fft_share": c.fft_seconds / dt,
            "sample": "%d passes of the same %dx%dx%d RVE; reference loop nests in C/OpenMP (oracle/c) + "
                      "pocketfft rfftn/irfftn with %d workers standing in for threaded FFTW" % (it, *n, c.threads)}


def slab_section(args, n, K, rank, world, local_rank, dist, torch):
    """The same RVE as ONE problem, x-slab decomposed over the ranks."""
    from fibergen_amd.distributed import DistributedLSSolver
    from fibergen_amd.rve import synthetic_fiber_rve
    from helpers import INCLUSION, MATRIX, lame
    if n[0] % world or n[1] % world:
        return {"error": "grid not divisible by the number of ranks"}
    scale = max(args.n, 128) / 128.0
    phi, normals = synthetic_fiber_rve(n, K=K, R=0.05 / scale, L=0.4 / scale, seed=0,
                                       with_normals=(args.mixing == "laminate"))
    s = DistributedLSSolver(*n, device=local_rank)
    s.set_num_phases(2)
    mats = [lame(**MATRIX), lame(**INCLUSION)]
    s.set_phase(0, mats[0][0], mats[0][1], s.slab(1.0 - phi))
    s.set_phase(1, mats[1][0], mats[1][1], s.slab(phi))
    if normals is not None:
        s.set_normals(s.slab(normals))
    s.set_options(mixing_rule=args.mixing)
    if os.environ.get("FG_SLAB_FUSE_X"):
        s.set_options(fuse_x=int(os.environ["FG_SLAB_FUSE_X"]))
    del phi, normals
    s.calc_ref_material()
    E = np.array([1.0, 0, 0, 0, 0, 0])
    s.iterate(E, max(2, args.warmup))
    torch.cuda.synchronize()
    dist.barrier()
    s.comm_time = 0.0
    t0 = time.perf_counter()
    s.iterate(E, args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, s.comm_time], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, comm = float(t[0].item()), float(t[1].item())
    return {"value": args.steps / dt, "unit": "it/s", "ms_per_step": 1e3 * dt / args.steps, "scaling": "strong",
            "exchange_ms_per_step": 1e3 * comm / args.steps,
            "parallelism": "x-slabs x%d, 2 all-to-all + 2 halo exchanges per pass (RCCL p2p)" % world}


def guarded_slab_section(args, n, K, rank, world, local_rank, dist, torch, out):
    import threading

    def give_up():
        if rank == 0 and out is not None:
            out["slab"] = {"error": "slab section exceeded %d s" % args.slab_timeout}
            print(json.dumps(out), flush=True)
        os._exit(0)
    timer = threading.Timer(args.slab_timeout, give_up)
    timer.daemon = True
    timer.start()
    try:
        return slab_section(args, n, K, rank, world, local_rank, dist, torch)
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        timer.cancel()


# HIP-event slot name -> kernel name in the rocprofv3 counter summaries under profiles/
PMC_KERNEL = {"u_eps_stress_div": ("k_u_tile", "k_u_fast"), "u_eps_stress_div_r2cz": "k_u_fast_z", "stress_div": "k_stress_div_voigt", "xfft_g0_xifft": ("k_xfused_persistent", "k_xfused"),
              "eps_norm": "k_eps_norm", "stress": "k_stress", "div": "k_div", "g0": "k_g0"}


def committed_traffic(n, slot):
    """HBM bytes per launch of kernel `slot` from the committed PMC summary of this workload
    (profiles/*pmc_hbm_traffic_<n>cubed*.csv: separate FETCH_SIZE / WRITE_SIZE passes of this same
    bench command, FETCH_SIZE doubled for gfx950).  A counter pass cannot run inside the timed
    process, so the figure is read back from the file; None when no pass exists for this grid."""
    import csv
    import glob
    want = PMC_KERNEL.get(slot)
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                          "*pmc_hbm_traffic_%dcubed*.csv" % n)))  # latest by name
    if not want or not files:
        return None, None
    rows = list(csv.DictReader(open(files[-1])))
    for w in (want if isinstance(want, tuple) else (want,)):
        for row in rows:
            if row["kernel"].split("<")[0] == w:
                return float(row["total_GB"]) * 1e9, "profiles/" + os.path.basename(files[-1])
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=256, help="grid size per axis (128, 256, 512 are the BASELINE sizes; any size runs)")
    ap.add_argument("--mixing", default="voigt", choices=["voigt", "laminate"])
    ap.add_argument("--mode", default="elasticity", choices=["elasticity", "porous", "heat", "viscosity"],
                    help="porous / heat: scalar potential, 3-component gradient; viscosity: dual Stokes scheme "
                         "(BASELINE config 5, 256^3)")
    ap.add_argument("--u-tile", type=int, default=None, help="override the solver's u_tile option (0, 8, 16)")
    ap.add_argument("--fuse-z", type=int, default=None, help="override the solver's fuse_z option (attach the z r2c to the sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-slab", action="store_true", help="N > 1: skip the slab-decomposed measurement")
    ap.add_argument("--slab-timeout", type=int, default=240)
    ap.add_argument("--force-slab", action="store_true",
                    help="also at N = 1: run the slab-decomposed driver (one slab, exchanges become local copies)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # torch is plumbing for the multi-rank barrier / max-reduction only.  It must be imported
    # BEFORE libfibergen_amd.so is loaded so that both share one HIP runtime (same soname).
    torch = None
    dist = None
    if world > 1 or args.force_slab:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from fibergen_amd import LSSolver
    from fibergen_amd.rve import synthetic_fiber_rve
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import INCLUSION, MATRIX, lame

    n = (args.n, args.n, args.n)
    scale = max(args.n, 128) / 128.0
    K = int(round(40 * scale ** 3)) if args.n >= 128 else 5
    phi, normals = synthetic_fiber_rve(n, K=K, R=0.05 / scale, L=0.4 / scale, seed=0,
                                       with_normals=(args.mixing == "laminate"))
    scalar = args.mode in ("porous", "heat")
    stokes = args.mode == "viscosity"
    s = LSSolver(*n, device=local_rank)
    if scalar or stokes:
        s.set_options(mode=args.mode)
    s.set_num_phases(2)
    mats = [(1.0, 0.0), (10.0, 0.0)] if scalar else [lame(**MATRIX), lame(**INCLUSION)]  # contrast 10
    if stokes:
        mats = [(1.0, 0.0), (0.1, 0.0)]   # fluid with ten times more viscous particles (fluidity constants)
    s.set_phase(0, mats[0][0], mats[0][1], 1.0 - phi)
    s.set_phase(1, mats[1][0], mats[1][1], phi)
    if normals is not None:
        s.set_normals(normals)
    s.set_options(mixing_rule=args.mixing)
    if args.u_tile is not None:
        s.set_options(u_tile=args.u_tile)
    if args.fuse_z is not None:
        s.set_options(fuse_z=args.fuse_z)
    vf = float(phi.mean())
    del phi, normals
    s.calc_ref_material()
    # each rank its own load case (calc_effective_properties' unit strains / gradients)
    E = np.zeros(3 if scalar else 6)
    E[rank % E.size] = 1.0
    if stokes:   # traceless prescribed stresses (F:26257-26261)
        E = [np.array([1.0, -1, 0, 0, 0, 0]), np.array([0, 1.0, -1, 0, 0, 0]), np.array([0, 0, 0, 1.0, 0, 0]),
             np.array([0, 0, 0, 0, 1.0, 0]), np.array([0, 0, 0, 0, 0, 1.0])][rank % 5]

    def sync():
        s.synchronize()  # the solver's own HIP stream carries all the work
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    s.iterate(E, args.warmup)
    sync()
    t0 = time.perf_counter()
    s.iterate(E, args.steps)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = world * args.steps / dt

    out = None
    if rank == 0:
        # per-kernel durations, measured live with HIP events on the solver's stream
        s.enable_stage_timing(True)
        s.iterate(E, min(args.steps, 20))
        times, cnt = s.stage_times()
        s.enable_stage_timing(False)
        ab = algorithmic_bytes(n, 2, args.mixing)
        if scalar:
            # one component through the FFT chain; the sweep reads T and phi and writes f
            F = n[0] * n[1] * (n[2] // 2 + 1)
            ab = {k: 32 * F for k in ab}
            ab["stress"] = 24 * n[0] * n[1] * n[2]
        kern = {}
        for k, ms in times.items():
            avg = ms / max(cnt, 1)
            if avg <= 0:
                continue   # stage absorbed by a fused kernel
            name, alg = k, ab[k]
            if scalar and k == "stress":
                name = "T_grad_flux_div"
            elif k == "stress" and times["div"] == 0 and times["eps_norm"] == 0:
                # displacement-based sweep: strain operator + polarisation + divergence + norms,
                # 3 u + phi in, 3 f out
                name, alg = "u_eps_stress_div", 56 * n[0] * n[1] * n[2]
                if times["r2c_z"] == 0:
                    # ... with the z r2c attached: the half spectrum of f goes out instead of f (3 x 16 B x nzf/nz)
                    name = "u_eps_stress_div_r2cz"
                    alg = (32 * n[2] + 48 * (n[2] // 2 + 1)) * n[0] * n[1]
            elif k == "stress" and times["div"] == 0:
                # polarisation + divergence in one sweep: 6 eps + phi in, 3 f out (SURVEY 8d "S + div: 80")
                name, alg = "stress_div", 80 * n[0] * n[1] * n[2]
            if k == "g0" and times["c2c_x_fwd"] == 0:
                # x-FFT, Green operator, inverse x-FFT: 3 complex components in, 3 out (48 B/voxel)
                name = "xfft_g0_xifft"
            kern[name] = {"avg_ms": avg, "alg_GB": alg / 1e9, "GBps": (alg / 1e9) / (avg / 1e3)}
        g0_alone = None
        if not scalar and not stokes:
            # north_star names the Green-operator apply on its own (">= 50 % of the HBM roofline in the
            # Gamma0-apply kernel"): in the default pipeline it is fused into the x pass, so time the
            # stand-alone kernel of the one-kernel-per-routine pipeline as well (96 B per frequency)
            s.set_options(fuse_x=0)
            s.enable_stage_timing(True)
            s.iterate(E, 5)
            t2, c2 = s.stage_times()   # accumulators continue from the measurement above
            s.enable_stage_timing(False)
            s.set_options(fuse_x=1)
            if c2 > cnt and t2["g0"] > times["g0"]:
                ms = (t2["g0"] - times["g0"]) / (c2 - cnt)
                g0_alone = {"kernel": "k_g0 (fuse_x=0)", "avg_ms": ms, "alg_GB": ab["g0"] / 1e9,
                            "GBps": ab["g0"] / 1e9 / (ms / 1e3), "frac_of_hbm_peak": ab["g0"] / 1e9 / (ms / 1e3) / HBM_PEAK_GBS}
        dom = max(kern, key=lambda k: kern[k]["avg_ms"])
        N = n[0] * n[1] * n[2]
        traffic, traffic_src = committed_traffic(args.n, dom)
        roof = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kern[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": int(kern[dom]["alg_GB"] * 1e9), "avg_launch_ms": kern[dom]["avg_ms"]}
        out = {
            "metric": "LS iterations/sec (basic scheme, staggered grid, %s)" % (("linear elastic" if not stokes else "Stokes flow, dual scheme") if not scalar else
                                                                               args.mode + " scalar mode"),
            "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 two-phase fibre RVE (K=%d capsules, vf=%.3f), contrast 10, mixing=%s, "
                                   "one load case per GPU" % (args.n, K, vf, args.mixing),
                       "grid": list(n), "mixing_rule": args.mixing, "mode": args.mode,
                       "parallelism": "loadcase x%d" % world},
            "roofline": roof,
            "loop_GBps_Amin": A_MIN_BYTES_PER_VOXEL * N * (args.steps / dt) / 1e9,
            "loop_GBps_Astage": A_STAGE_BYTES_PER_VOXEL * N * (args.steps / dt) / 1e9,
            "kernels": kern,
        }
        if g0_alone is not None:
            out["gamma0_apply_standalone"] = g0_alone
        # the measured roofline: streaming copy / triad of the library on this GPU, 1 GiB arrays
        try:
            import ctypes
            from fibergen_amd import _lib
            cg, tg = ctypes.c_double(0.0), ctypes.c_double(0.0)
            if _lib.load().fg_hbm_stream(local_rank, 1024, 3, ctypes.byref(cg), ctypes.byref(tg)) == 0 and tg.value > 0:
                best = max(cg.value, tg.value)
                out["hbm_stream"] = {"copy_GBps": cg.value, "triad_GBps": tg.value,
                                     "dominant_kernel_frac_of_measured": kern[dom]["GBps"] / best}
                if g0_alone is not None:
                    out["hbm_stream"]["gamma0_apply_frac_of_measured"] = g0_alone["GBps"] / best
        except Exception as e:  # noqa: BLE001
            out["hbm_stream"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if scalar:
            # the loop-level figures use the elasticity byte counts; per voxel the scalar loop moves
            # 24 (sweep) + 7 x 16 (six 1-component FFT passes + Green operator) bytes
            per = 24 + 7 * 16
            out["loop_GBps_Amin"] = per * N * (args.steps / dt) / 1e9
            out["loop_GBps_Astage"] = out["loop_GBps_Amin"]
        if world == 1 and not args.no_cpu_baseline and not scalar and not stokes:
            s.close()
            out["cpu_baseline"] = cpu_baseline(n, args.mixing, args.cpu_budget)
    if (world > 1 or args.force_slab) and not args.no_slab and not scalar and not stokes:
        # Second measurement: ONE problem slab-decomposed over all ranks (x-slabs, two RCCL
        # all-to-alls per pass).  Guarded: a failure or a stall here must not cost the line above.
        s.close()
        slab = guarded_slab_section(args, n, K, rank, world, local_rank, dist, torch, out)
        if out is not None:
            out["slab"] = slab
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
