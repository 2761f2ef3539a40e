#!/usr/bin/env python3
"""Benchmark of the Lippmann-Schwinger basic-scheme pass on MI355X.

    python bench.py --gpus N --steps K --warmup W [--n 256] [--mixing voigt|laminate]

A "step" is one LS iteration (one basicScheme pass, F:20558-20578) on a synthetic fibre RVE resident in HBM
(non-overlapping capsules from the build's seeded placer).

  N = 1   the single-GPU displacement loop.  The K-step region is timed `--repeats` times, `ms_per_step` is the
          median; `sustained_it_s` is a >= 2 s run; `run_load_case_it_s` times fg_run_load_case (stop rule included);
          `also` carries the other BASELINE sizes (128^3 Voigt, 512^3 laminate = the north-star target config);
          `slab_forced` the same problem through the slab driver as ONE slab; `cpu_baseline` the reference's loop in
          C/OpenMP on the host cores (thread sweep + the reference's default of one thread).
  N > 1   one rank per GPU: ONE problem, x-slab decomposed over the ranks, RCCL all-to-all between the FFT axes: `value`
          is its it/s ("scaling": "strong"); `replicas` = every rank its own load case of the same RVE (no collective,
          the six load cases of calc_effective_properties are independent).  Started under torchrun (WORLD_SIZE set) the
          process is one rank; started plainly (`python bench.py --gpus 8`) it launches the N ranks itself -- before it
          has touched the GPU -- relays rank 0's line and exits with the ranks' code.  The line names the ranks' devices
          (PCI bus ids), the transport and the time the exchanges take (`alltoall_ms`).

Rank 0 prints one JSON line (metric of BASELINE.json: LS iterations/s, plus the HBM roofline of the dominant kernel).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
A_MIN_BYTES_PER_VOXEL = 392   # maximum legal fusion with the strain as the state (kept so that rounds compare)


def algorithmic_bytes(n, nphases):
    """Algorithmic HBM bytes per launch of each kernel (DESIGN.md / SURVEY 8d): inputs read once, outputs written
    once, float64."""
    nx, ny, nz = n
    N = nx * ny * nz
    F = nx * ny * (nz // 2 + 1)   # complex frequencies
    stress = (6 + 1 + 6) * 8 * N if nphases == 2 else (12 + nphases) * 8 * N
    fft_pass = 3 * 2 * 16 * F     # 3 components, read + write, complex128
    return {"stress": stress, "div": 9 * 8 * N, "r2c_z": fft_pass, "c2c_y_fwd": fft_pass, "c2c_x_fwd": fft_pass,
            "g0": 96 * F, "c2c_x_inv": fft_pass, "c2c_y_inv": fft_pass, "c2r_z": fft_pass, "eps_norm": 9 * 8 * N}


def materials(mode):
    from helpers import INCLUSION, MATRIX, lame
    if mode in ("porous", "heat"):
        return [(1.0, 0.0), (10.0, 0.0)]
    if mode == "viscosity":
        return [(1.0, 0.0), (0.1, 0.0)]   # fluid with ten times more viscous particles (fluidity constants)
    return [lame(**MATRIX), lame(**INCLUSION)]   # contrast 10


def configure(s, phi, normals, mixing, mode, slab=None):
    """materials, phases, normals, options on an LSSolver / SlabGroup / DistributedLSSolver"""
    if mode != "elasticity":
        s.set_options(mode=mode)
    s.set_num_phases(2)
    mats = materials(mode)
    cut = (lambda a: a) if slab is None else slab
    s.set_phase(0, mats[0][0], mats[0][1], cut(1.0 - phi))
    s.set_phase(1, mats[1][0], mats[1][1], cut(phi))
    if normals is not None:
        s.set_normals(cut(normals))
    s.set_options(mixing_rule=mixing)


def timed_regions(iterate, sync, steps, warmup, repeats):
    """W warm-up steps, then `repeats` regions of exactly `steps` steps, each bracketed by sync() on both sides."""
    iterate(warmup)
    sync()
    out = []
    for _ in range(repeats):
        sync()
        t0 = time.perf_counter()
        iterate(steps)
        sync()
        out.append(time.perf_counter() - t0)
    return out


def kernel_table(s, E, n, steps, scalar, stokes=False):
    """per-kernel durations, measured live with HIP events on the solver's stream -> {name: avg_ms, alg_GB, GBps}"""
    s.enable_stage_timing(True)
    s.iterate(E, steps)
    times, cnt = s.stage_times()
    s.enable_stage_timing(False)
    kernel_table.last_sum_ms = sum(times.values()) / max(cnt, 1)
    kernel_table.last_bias_ms = s.stage_timing_bias()
    N = n[0] * n[1] * n[2]
    ab = algorithmic_bytes(n, 2)
    if scalar:   # one component through the FFT chain; the sweep reads T and phi and writes f
        F = n[0] * n[1] * (n[2] // 2 + 1)
        ab = {k: 32 * F for k in ab}
        ab["stress"] = 24 * N
    chunked = s.counter("pair_chunk_planes") > 0
    kern = {}
    for k, ms in times.items():
        avg = ms / max(cnt, 1)
        if avg <= 0:
            continue   # stage absorbed by a fused kernel
        name, alg = k, ab[k]
        if scalar and k == "stress":
            name = "T_grad_flux_div"
        elif k == "stress" and times["div"] == 0 and times["eps_norm"] == 0:
            # displacement sweep: strain operator + polarisation + divergence + norms, 3 u + phi in, 3 f out
            name, alg = "u_eps_stress_div", 56 * N
        elif k == "stress" and times["div"] == 0:
            name, alg = "stress_div", 80 * N   # SURVEY 8d "S + div: 80"
        if stokes and k == "eps_norm":
            # viscosity: the slot is the Delta-operator tail k_eps_delta (F:20438-20452): strain from u, the polarisation
            # re-evaluated from the old stress field and the two phase fractions, eta written, norms: 3 u + 6 old + 2 phi in,
            # 6 out = 136 B per voxel (rounds 1-4 priced it as the elastic strain sweep, 72 B)
            name, alg = "eps_delta_norm", (3 + 6 + 2 + 6) * 8 * N
        if k == "g0" and times["c2c_x_fwd"] == 0:
            name = "xfft_g0_xifft"             # x-FFT, Green operator, inverse x-FFT: 48 B/voxel
        if k == "r2c_z" and times["c2c_y_fwd"] == 0 and n[1] > 1:
            name = "zy_plane_fwd"              # plane kernel: r2c along z + c2c along y in one pass (fg_fft_plane.h)
            if chunked:   # r2c(c) -> y(c) in runs of x planes (option pair_chunk): the slot holds both passes of every chunk
                name, alg = "r2c_z+c2c_y_fwd_chunked", ab["r2c_z"] + ab["c2c_y_fwd"]
        if k == "c2r_z" and times["c2c_y_inv"] == 0 and n[1] > 1:
            name = "yz_plane_inv"
        if k == "c2c_y_inv" and times["c2r_z"] == 0 and n[1] > 1 and chunked:
            name, alg = "c2c_y_inv+c2r_z_chunked", ab["c2c_y_inv"] + ab["c2r_z"]
        kern[name] = {"avg_ms": avg, "alg_GB": alg / 1e9, "GBps": (alg / 1e9) / (avg / 1e3)}
    return kern, times, cnt


# HIP-event slot name -> kernel names in the rocprofv3 counter summaries under profiles/
PMC_KERNEL = {"u_eps_stress_div": ("k_u_tile", "k_u_fast"),
              "stress_div": ("k_stress_div_voigt", "k_eps_tile"), "xfft_g0_xifft": ("k_xfused",), "eps_norm": ("k_eps_norm",),
              "eps_delta_norm": ("k_eps_delta",),
              "stress": ("k_stress",), "div": ("k_div",), "g0": ("k_g0",), "r2c_z": ("k_zpass<fg::fft::R2CKernel",),
              "c2r_z": ("k_zpass<fg::fft::C2RKernel",), "c2c_y_fwd": ("k_strided<fg::fft::StridedKernel<*, -1>",),
              "c2c_y_inv": ("k_strided<fg::fft::StridedKernel<*, 1>",), "zy_plane_fwd": ("k_plane<fg::fft::ZYKernel",),
              "yz_plane_inv": ("k_plane<fg::fft::YZKernel",)}


def committed_traffic(n, mixing, slot, default_options):
    """HBM bytes per launch of kernel `slot` from the newest committed PMC summary of this workload
    (profiles/rNN_pmc_hbm_traffic_<n>cubed[_<mixing>]_vM.csv: separate FETCH_SIZE / WRITE_SIZE passes of this same bench
    command, gfx950 corrections applied).  A counter pass cannot run inside the timed process, so the figure is read
    back from the file (`traffic_source` says which); None when the run deviates from the default options or no pass
    exists for this grid."""
    import csv
    import glob
    import re
    if not default_options or slot not in PMC_KERNEL:
        return None, None

    def version(path):
        m = re.match(r"r(\d+)_.*?_v(\d+)", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic_%dcubed*.csv" % n))
             if ("laminate" in os.path.basename(f)) == (mixing == "laminate")]
    if not files:
        return None, None
    newest = max(files, key=version)
    rows = list(csv.DictReader(open(newest)))
    def matches(name, w):   # "k_x": the name before its template list; "k_x<...": a prefix, `*` = any run of characters
        if "<" not in w:
            return name.split("<")[0] == w
        head, _, tail = w.partition("*")
        return name.startswith(head) and (not tail or tail in name[len(head):])
    for w in PMC_KERNEL[slot]:
        for row in rows:
            if matches(row["kernel"], w):
                return float(row["total_GB"]) * 1e9, "profiles/" + os.path.basename(newest)
    return None, None


def live_traffic(args, slot):
    """HBM bytes per launch of kernel `slot`, measured NOW on this box: two short child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only, as MI355X_MICROARCH.md's HBM section
    prescribes; the child does a handful of passes of the same workload and nothing else), FETCH_SIZE doubled for gfx950, KB ->
    bytes, averaged over the launches of the kernel.  -> (bytes, {"fetch": .., "write": .., "launches": ..}) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if slot not in PMC_KERNEL:
        return None, "no kernel name known for slot %s" % slot
    prof = shutil.which("rocprofv3")
    if not prof:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler: no nested counter passes"

    def matches(name, w):   # "k_x": the name before its template list; "k_x<...": a prefix, `*` = any run of characters
        name = name[name.index("k_"):] if "k_" in name else name
        if "<" not in w:
            return name.split("<")[0].split("(")[0] == w
        head, _, tail = w.partition("*")
        return name.startswith(head) and (not tail or tail in name[len(head):])
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="fg_pmc_")
        try:
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-child", "--n", str(args.n), "--mixing", args.mixing, "--mode", args.mode]
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240,
                               env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
            vals = []
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == counter and any(matches(r["Kernel_Name"], w) for w in PMC_KERNEL[slot]):
                        vals.append(float(r["Counter_Value"]))
            if p.returncode != 0 or not vals:
                return None, "%s pass: rc %d, %d launches of the kernel seen (%s)" % (
                    counter, p.returncode, len(vals), (p.stderr.strip().splitlines() or [""])[-1][:160])
            got[counter] = (sum(vals) / len(vals), len(vals))
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, "%s pass: %s" % (counter, type(e).__name__)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fetch = 2 * got["FETCH_SIZE"][0] * 1024
    write = got["WRITE_SIZE"][0] * 1024
    return fetch + write, {"fetch_bytes": fetch, "write_bytes": write, "launches_averaged": got["FETCH_SIZE"][1],
                           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate child runs of this script on this "
                                     "box; FETCH_SIZE x 2 (gfx950), KB -> bytes"}


def traffic_child(args):
    """the workload of live_traffic's child runs: a handful of passes, nothing else"""
    from fibergen_amd import LSSolver
    from fibergen_amd.rve import bench_rve
    phi, normals, _ = bench_rve(args.n, args.mixing)
    s = LSSolver(args.n, args.n, args.n)
    configure(s, phi, normals, args.mixing, args.mode)
    s.calc_ref_material()
    E = np.zeros(3 if args.mode in ("porous", "heat") else 6)
    E[0] = 1.0
    if args.mode == "viscosity":
        E = np.array([1.0, -1, 0, 0, 0, 0])
    s.iterate(E, 6)
    s.synchronize()
    s.close()


def physical_cores():
    """(physical cores, sockets) of this host from /proc/cpuinfo; (None, None) when it cannot be read"""
    try:
        cores, phys = set(), set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                    phys.add(pid)
                pid = cid = None
        return (len(cores) or None), (len(phys) or None)
    except OSError:
        return None, None


A_STAGE_BYTES_PER_VOXEL = 632   # the reference's pass structure, every stage reading its inputs and writing its outputs once


def cpu_limits():
    """What the node lets this process use: CPUs in the affinity mask, the CFS bandwidth quota of its control group (cgroup v2
    cpu.max or v1 cpu.cfs_quota_us / cpu.cfs_period_us) in CPUs, and the throttling counter to take differences of.  A thread
    sweep that stops scaling at 16 threads on a 128-core host is what a quota of about that many CPUs looks like from inside."""
    out = {"affinity_cpus": None, "cgroup_quota_cpus": None}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["cgroup_quota_cpus"] = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cgroup_quota_cpus"] = None if q <= 0 else q / per
        except (OSError, ValueError):
            pass
    return out


def cpu_throttled():
    """(periods throttled, seconds throttled) of this control group so far, or None"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            kv = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
            n = int(kv.get("nr_throttled", 0))
            t = float(kv["throttled_usec"]) / 1e6 if "throttled_usec" in kv else float(kv.get("throttled_time", 0)) / 1e9
            return n, t
        except (OSError, ValueError):
            continue
    return None


def cpu_passes(grid, mixing, phi_path, normals_path, threads, max_passes, max_seconds, bind=True, loops="reference"):
    """One run of the CPU stand-in (oracle/cpu_loop.py) in a process of its own with `threads` OpenMP threads, pinned one per
    core and spread over the sockets (OMP_PROC_BIND=spread, OMP_PLACES=cores: every thread of a pass is an OpenMP thread since
    the transforms are oracle/c's own), buffers first touched by the threads that sweep them.
    -> (it/s, fft share of the time, passes timed)"""
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), FG_REF_NATIVE_BUILT="1")
    if bind:
        env.update(OMP_PROC_BIND="spread", OMP_PLACES="cores")
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_loop.py"), "--grid", str(grid), "--mixing", mixing, "--phi", phi_path,
           "--threads", str(threads), "--max-passes", str(max_passes), "--max-seconds", str(max_seconds), "--loops", loops]
    if normals_path:
        cmd += ["--normals", normals_path]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        raise RuntimeError("oracle/cpu_loop.py failed (%d): %s" % (p.returncode, p.stderr.strip().splitlines()[-1:] or ""))
    r = json.loads(lines[-1])
    return r["it_s"], r["fft_share"], r["passes"]


def cpu_baseline(n, mixing, phi, normals, budget_s=25.0, others=()):
    """The reference's loop restated in C/OpenMP (oracle/c: one in-place strain field, buffers allocated once, the
    reference's release flags -O3 -march=native built on this host, its own threaded row-column FFT in place of threaded
    FFTW) on the host cores: a sweep over thread counts on the headline grid (best = `value`), the reference's default of
    ONE thread (F:25226), and the other BASELINE grids `others` = [(n_edge, mixing, phi, normals)] at the best thread
    count (`per_grid`).  Every count runs in a fresh process, threads pinned one per core and spread over the sockets."""
    import shutil
    import tempfile
    from oracle.c_oracle import load
    load(native=True)   # -O3 -march=native, built once on this host; the runs below reuse it
    ncpu = os.cpu_count() or 1
    cores, sockets = physical_cores()
    per_socket = (cores // sockets) if cores and sockets else None
    limits = cpu_limits()
    usable = min(x for x in (ncpu, limits["affinity_cpus"], limits["cgroup_quota_cpus"]) if x)
    # thread counts up to what the node grants (a CFS quota of 16 CPUs makes 32 threads slower than 16: throttling, not the code)
    counts = sorted({t for t in (4, 8, 16, 32, per_socket or 64, cores or ncpu, int(usable)) if t and t <= usable}) or [1]
    thr0 = cpu_throttled()
    t_start = time.perf_counter()
    # inputs are handed to the child processes as files: memory-backed where /dev/shm has the room (a container's default is
    # 64 MB; the 512^3 laminate leg stages 4 GB), the default temporary directory otherwise
    biggest = max([n[0]] + [o[0] for o in others])
    need_stage = 1.05 * 8 * biggest ** 3 * 4
    stage_dir = None
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need_stage:
            stage_dir = "/dev/shm"
    except OSError:
        stage_dir = None
    tmp = tempfile.mkdtemp(prefix="fg_cpu_", dir=stage_dir)

    def stash(tag, ph, nr):
        pp = os.path.join(tmp, tag + "_phi.npy")
        np.save(pp, np.ascontiguousarray(ph, dtype=np.float64))
        npth = ""
        if nr is not None:
            npth = os.path.join(tmp, tag + "_normals.npy")
            np.save(npth, np.ascontiguousarray(nr, dtype=np.float64))
        return pp, npth
    try:
        pp, npth = stash("head", phi, normals)
        sweep, shares = {}, {}
        for th in counts:
            sweep[th], shares[th], _ = cpu_passes(n[0], mixing, pp, npth, th, 20, min(3.0, budget_s / (2 * len(counts))))
        best = max(sweep, key=sweep.get)
        one = cpu_passes(n[0], mixing, pp, npth, 1, 1, 0.0)[0] if n[0] <= 256 else None
        # what the reference's traversal orders cost: the same sweep with the two stencil operators z-innermost (same values)
        tuned = {}
        for th in counts:
            try:
                tuned[th] = cpu_passes(n[0], mixing, pp, npth, th, 20, min(2.0, budget_s / (3 * len(counts))), loops="contiguous")[0]
            except Exception:  # noqa: BLE001
                break
        per_grid = {"%d^3 %s" % (n[0], mixing): {"it_s": sweep[best], "threads": int(best), "fft_share": shares[best]}}
        for ne, mix, ph, nr in others:
            key = "%d^3 %s" % (ne, mix)
            try:
                need = 34 * ne ** 3 * 8 * 1.1   # strain 6 + work 3 + spectrum 3 + phases 2 + normals 3, twice the inputs (files + placed copies)
                avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
                if need > 0.8 * avail:
                    per_grid[key] = {"skipped": "needs %.0f GB of host memory, %.0f GB available" % (need / 1e9, avail / 1e9)}
                    continue
                p2, n2 = stash("g%d%s" % (ne, mix), ph, nr)
                v, sh, it = cpu_passes(ne, mix, p2, n2, best, 20 if ne < 256 else 3, 3.0)
                per_grid[key] = {"it_s": v, "threads": int(best), "fft_share": sh, "passes": it}
                for f in (p2, n2):
                    if f:
                        os.remove(f)
            except Exception as e:  # noqa: BLE001
                per_grid[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    ordered = [sweep[k] for k in sorted(sweep)]
    monotone = all(b >= 0.97 * a for a, b in zip(ordered, ordered[1:]))
    thr1 = cpu_throttled()
    limits["usable_cpus"] = usable
    if thr0 and thr1:
        limits["cfs_throttled_periods_during_sweep"] = thr1[0] - thr0[0]
        limits["cfs_throttled_seconds_during_sweep"] = round(thr1[1] - thr0[1], 3)
    N = n[0] * n[1] * n[2]
    return {"value": sweep[best], "unit": "it/s", "cores": int(best), "kind": "port",
            "one_thread_it_s": one, "host_cpus": ncpu, "physical_cores": cores, "sockets": sockets, "fft_share": shares[best],
            "thread_sweep_it_s": {str(k): v for k, v in sorted(sweep.items())}, "per_grid": per_grid,
            "cpu_GBps": A_STAGE_BYTES_PER_VOXEL * N * sweep[best] / 1e9,
            "scales_with_threads": monotone, "cpu_limits": limits,
            "tuned_loops_sweep_it_s": {str(k): v for k, v in sorted(tuned.items())},
            "tuned_loops_note": "NOT the baseline: the same pass with divOperatorStaggered / epsOperatorStaggered restated z-innermost "
                                "(oracle/c ref_div_contig / ref_eps_contig, bit-identical values).  The reference runs their x- and y-"
                                "difference nests z-OUTERMOST (F:18864-18887, F:18646-18675): strided inner loops, and neighbouring z -- "
                                "one cache line -- on different threads; `value` keeps those orders because they are the reference's "
                                "OpenMP path",
            "note": ("the node grants this process %g CPUs (affinity %s, cgroup quota %s of %d logical CPUs): the sweep stops there; "
                     % (usable, limits["affinity_cpus"], limits["cgroup_quota_cpus"], ncpu)) +
                    ("it/s grows with the thread count over the sweep" if monotone else
                     "does not scale over the whole sweep: best at %d of %s threads" % (best, "/".join(str(k) for k in counts))) +
                    "; cpu_GBps prices a pass at the reference's 632 B/voxel (SURVEY 8d)",
            "sample": "passes of the same %dx%dx%d RVE for <= 3 s per thread count (%s threads; value = best; every count in a "
                      "fresh process: OMP_PROC_BIND=spread OMP_PLACES=cores, buffers first touched by the threads that sweep "
                      "them) + one pass with 1 thread; per_grid: the other BASELINE grids at the best count (512^3: 3 passes); "
                      "oracle/c loop nests (-O3 -march=native, in-place strain field, preallocated) + oracle/c's own threaded "
                      "row-column FFT (radix-4 Stockham lines, 16 at a time) standing in for threaded FFTW; %.0f s in total"
                      % (*n, "/".join(str(k) for k in counts), time.perf_counter() - t_start)}


def cg_rate(s, E, steps, repeats, sync):
    """CG iterations per second through the solver's run entry point (stop rule every iteration, maxiter = steps)"""
    s.set_options(method="cg", tol=0.0, abs_tol=0.0, maxiter=steps)
    rates = []
    for _ in range(repeats + 1):
        sync()
        t0 = time.perf_counter()
        s.run(E)
        sync()
        rates.append((s.iterations + 1) / (time.perf_counter() - t0))   # iterations 0 .. maxiter
    s.set_options(method="basic", tol=1e-4, abs_tol=np.finfo(float).eps, maxiter=10000)
    return statistics.median(rates[1:]), min(rates[1:]), max(rates[1:])


def cache_stream(device):
    """Streaming copy on arrays that stay in the 256 MB Infinity Cache (2 x 48 MB): the ceiling of a pass whose fields are
    cache resident (128^3: three components = 51 MB), where the 8 TB/s HBM peak is the wrong yardstick"""
    try:
        import ctypes
        from fibergen_amd import _lib
        cg, tg = ctypes.c_double(0.0), ctypes.c_double(0.0)
        if _lib.load().fg_hbm_stream(int(device), 48, 10, ctypes.byref(cg), ctypes.byref(tg)) != 0:
            return None
        return {"copy_GBps": cg.value, "triad_GBps": tg.value, "array_MB": 48}
    except Exception:  # noqa: BLE001
        return None


def measure_single(args, n_edge, mixing, mode, device, E, detail):
    """One single-GPU workload: median-of-repeats it/s (+ kernels, sustained, run_load_case when detail)."""
    from fibergen_amd import LSSolver
    from fibergen_amd.rve import bench_rve
    n = (n_edge,) * 3
    phi, normals, par = bench_rve(n_edge, mixing)
    s = LSSolver(*n, device=device)
    configure(s, phi, normals, mixing, mode)
    if args.u_tile is not None:
        s.set_options(u_tile=args.u_tile)
    s.calc_ref_material()
    scalar = mode in ("porous", "heat")
    dts = timed_regions(lambda k: s.iterate(E, k), s.synchronize, args.steps, args.warmup, args.repeats)
    med = statistics.median(dts)
    res = {"it_s": args.steps / med, "ms_per_step": 1e3 * med / args.steps,
           "ms_per_step_min": 1e3 * min(dts) / args.steps, "ms_per_step_max": 1e3 * max(dts) / args.steps,
           "repeats": len(dts), "rve": {"K": par["K"], "R": float(par["R"]), "L": float(par["L"]), "vf": float(phi.mean()),
                                        "interface_voxel_fraction": float(((phi > 0) & (phi < 1)).mean())}}
    kern, times, cnt = kernel_table(s, E, n, min(args.steps, 20), scalar, mode == "viscosity")
    dom = max(kern, key=lambda k: kern[k]["avg_ms"])
    res["dominant_kernel"] = {"kernel": dom, **kern[dom], "frac_of_hbm_peak": kern[dom]["GBps"] / HBM_PEAK_GBS}
    res["kernels"] = kern
    # the per-kernel HIP-event figures (an empty event pair's reading subtracted) against the step they make up
    res["kernel_sum_ms"] = kernel_table.last_sum_ms
    res["hip_event_bias_ms_subtracted"] = kernel_table.last_bias_ms
    res["loop_alg_GBps"] = sum(v["alg_GB"] for v in kern.values()) / (res["ms_per_step"] / 1e3)
    if detail:
        # sustained clocks: >= sustain_s of back-to-back passes
        chunk = max(args.steps, int(0.25 / max(med / args.steps, 1e-6)))
        s.synchronize()
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < args.sustain_s:
            s.iterate(E, chunk)
            s.synchronize()
            done += chunk
        res["sustained_it_s"] = done / (time.perf_counter() - t0)
        res["sustained_s"] = time.perf_counter() - t0
        # the reference's own entry point: LSSolver::run with the stop rule (norms reach the host every pass); maxiter chosen
        # for ~0.3 s so that the fixed costs of a run (reference-material scan, final strain sweep) do not dominate
        n_run = max(args.steps, min(2000, int(0.3 / max(med / args.steps, 1e-6))))
        s.set_options(tol=0.0, abs_tol=0.0, maxiter=n_run)
        s.run(E)
        res["run_load_case_it_s"] = s.iterations / s.solve_time
        res["run_load_case_iterations"] = s.iterations
        s.set_options(tol=1e-4, abs_tol=np.finfo(float).eps, maxiter=10000)
        # The boundary takes host arrays (fg_set_phase) and returns host arrays (fg_get_field): one load case the way a caller
        # of the C ABI sees it -- phase fractions over PCIe in, the reference's default tolerance 1e-4, strain and stress
        # fields out.  Reported beside `value`, never as `value` (whose inputs are resident in HBM).
        mats = materials(mode)
        phi0 = 1.0 - phi   # (the caller's array: formed before the clock starts)
        t0 = time.perf_counter()
        s.set_phase(0, mats[0][0], mats[0][1], phi0)
        s.set_phase(1, mats[1][0], mats[1][1], phi)
        s.synchronize()
        t1 = time.perf_counter()
        s.run(E)
        s.synchronize()
        t2 = time.perf_counter()
        eps_h = s.get_field("epsilon")
        sig_h = s.get_field("sigma")
        t3 = time.perf_counter()
        res["pcie_inclusive"] = {"iterations": s.iterations, "tol": 1e-4, "upload_ms": 1e3 * (t1 - t0), "run_ms": 1e3 * (t2 - t1),
                                 "download_ms": 1e3 * (t3 - t2), "host_MB_in": 2 * phi.nbytes / 1e6,
                                 "host_MB_out": (eps_h.nbytes + sig_h.nbytes) / 1e6,
                                 "it_s_inclusive": s.iterations / (t3 - t0), "it_s_run_only": s.iterations / (t2 - t1)}
        del eps_h, sig_h
    return s, res, phi, normals


def cg_leg(args, s, E):
    """the reference's default method (runCGElasticity F:23153-23247) on the same workload: reported beside the basic scheme in
    every detailed line (`value` is its rate only with --method cg).  Run as the LAST measurement on the solver (it allocates the
    CG vectors and rewrites method / tol / maxiter), right before the solver is closed."""
    med_cg, lo_cg, hi_cg = cg_rate(s, E, args.steps, 5 if args.method == "cg" else 3, s.synchronize)
    return {"it_s": med_cg, "it_s_min": lo_cg, "it_s_max": hi_cg, "iterations_per_run": args.steps + 1}


def main():
    # dmabuf IPC (the host driver has no legacy IPC): read by the HSA runtime when the first HIP call initialises it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7, help="the K-step region is timed this many times; the median is reported")
    ap.add_argument("--sustain-s", type=float, default=10.0)
    ap.add_argument("--n", "--size", dest="n", type=int, default=256, help="grid size per axis (128, 256, 512 are the BASELINE sizes; any size runs)")
    ap.add_argument("--mixing", default="voigt", choices=["voigt", "laminate"])
    ap.add_argument("--mode", default="elasticity", choices=["elasticity", "porous", "heat", "viscosity"])
    ap.add_argument("--method", default="basic", choices=["basic", "cg"],
                    help="cg: `value` is the rate of runCGElasticity iterations (F:23153-23247, the reference's default method; one "
                         "operator application + two inner-product sweeps + two vector updates each) through fg_run_load_case "
                         "with maxiter = steps; the kernel table stays the basic scheme's (the operator is the same kernels)")
    ap.add_argument("--also", default="128:voigt,512:laminate,200:voigt,400:voigt,256:voigt:porous,256:voigt:viscosity",
                    help="N = 1: further single-GPU workloads n:mixing[:mode] reported under `also` ('' = none): the other "
                         "BASELINE sizes, two decimal sizes (200^3, 400^3: the transform passes of fg_fft_smooth.h) and config 5 "
                         "(porous / Stokes)")
    ap.add_argument("--u-tile", type=int, default=None, help="override the solver's u_tile option (0 = untiled sweep, 1 = tiled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not run the two rocprofv3 counter passes that measure roofline.traffic on this box (the figure "
                         "is then read from the newest committed PMC summary and flagged as such)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    ap.add_argument("--slab-members", type=int, default=1,
                    help="N = 1: also run the slab driver with this many slabs on the one GPU (0 = skip)")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1: skip the load-case replica measurement")
    ap.add_argument("--slab-timeout", type=int, default=300)
    ap.add_argument("--also-slab", default="128:voigt,512:laminate",
                    help="N > 1: further slab-decomposed workloads n:mixing[,n:mixing...] after the headline one, reported under `also_slab` "
                         "(default: BASELINE's north-star target configuration; '' = none).  Its failure or timeout leaves "
                         "the headline line untouched")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo", "nccl-one-gpu"],
                    help="N > 1: torch.distributed backend.  gloo is the dry run of this script's multi-rank path on a box with "
                         "ONE GPU: all ranks share device 0 and the slab exchanges are staged through the host.  nccl-one-gpu: "
                         "the real RCCL path (torch's process group AND the library's transport) with all ranks on device 0 -- "
                         "every rank poses as a host of its own (NCCL_HOSTID), RCCL connects them over loop-back sockets")
    # ranks started by this script's own launcher take their arguments from the environment (torchrun's argparse trips over
    # abbreviations such as --n before it hands the rest to the script)
    # -- only when the launcher's own marker names this rank's parent process chain: a stale FG_BENCH_ARGV in the environment
    # of a manual `torchrun bench.py ...` must not override that command line)
    launched = ("FG_BENCH_ARGV" in os.environ and "WORLD_SIZE" in os.environ and
                os.environ.get("FG_BENCH_LAUNCHED", "") == os.environ.get("TORCHELASTIC_RUN_ID", "?"))
    args = ap.parse_args(json.loads(os.environ["FG_BENCH_ARGV"]) if launched else None)

    if args.traffic_child:
        traffic_child(args)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly: launch the N ranks as fresh processes (torch.distributed.run, one rank per GPU) and relay
        # rank 0's line.  Nothing in this process has touched the GPU (numpy only so far), and nothing is exec'ed.
        # The rendezvous store picks its own free port (c10d endpoint :0) -- no bind-close-reuse window for another job
        # on a shared box to take the port in between.
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        run_id = "fgbench-%d-%d" % (os.getpid(), int(time.time()))
        env["FG_BENCH_ARGV"] = json.dumps(sys.argv[1:])
        env["FG_BENCH_LAUNCHED"] = run_id
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--rdzv-backend", "c10d", "--rdzv-endpoint", "127.0.0.1:0", "--rdzv-id", run_id, "--local-addr", "127.0.0.1",
               os.path.abspath(__file__), "--gpus", str(args.gpus)]
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        for ln in proc.stdout.splitlines():
            if not ln.startswith("{"):
                print(ln, file=sys.stderr)
        if lines:
            print(lines[-1], flush=True)
        elif proc.returncode == 0:
            proc.returncode = 6
        sys.exit(proc.returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE = %d: the launcher's rank count is used" % (args.gpus, world), file=sys.stderr)
    default_options = args.u_tile is None and args.mode == "elasticity"
    scalar = args.mode in ("porous", "heat")
    stokes = args.mode == "viscosity"
    n = (args.n,) * 3
    N = args.n ** 3
    metric = "LS iterations/sec (basic scheme, staggered grid, %s)" % (
        args.mode + " scalar mode" if scalar else ("Stokes flow, dual scheme" if stokes else "linear elastic"))
    if args.gpus == 1 and args.method != "cg":
        metric += "; value = fg_iterate passes/s, run_load_case_it_s = the same loop under the stop rule"

    if world == 1:
        # ------------------------------------------------------------------ one GPU
        E = np.zeros(3 if scalar else 6)
        E[0] = 1.0
        if stokes:
            E = np.array([1.0, -1, 0, 0, 0, 0])   # traceless prescribed stress (F:26257-26261)
        s, res, phi, normals = measure_single(args, args.n, args.mixing, args.mode, local_rank, E, detail=True)
        kern = res["kernels"]
        dom = res["dominant_kernel"]["kernel"]
        traffic, traffic_src = committed_traffic(args.n, args.mixing, dom, default_options)
        roof = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kern[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_from_committed_profile": traffic_src,
                "traffic_measured_in_this_run": False,
                "alg_bytes_per_launch": int(kern[dom]["alg_GB"] * 1e9), "avg_launch_ms": kern[dom]["avg_ms"]}
        it_s = res["it_s"]
        if args.method == "cg":
            metric = metric.replace("LS iterations/sec (basic scheme", "CG iterations/sec (runCGElasticity")
        per_voxel = (24 + 7 * 16) if scalar else A_MIN_BYTES_PER_VOXEL
        out = {
            "metric": metric, "value": it_s, "unit": "it/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": None, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # `value` times fg_iterate: K passes enqueued back to back, the stop rule's sums stay on the device.  The loop SURVEY
            # 8d defines -- fg_run_load_case, the norms reaching the host every pass -- is run_load_case_it_s, here and top level
            "config": {"workload": "%d^3 two-phase fibre RVE (K=%d non-overlapping capsules, R=%.4f, L=%.3f, vf=%.3f), contrast 10, "
                                   "mixing=%s" % (args.n, res["rve"]["K"], res["rve"]["R"], res["rve"]["L"], res["rve"]["vf"], args.mixing),
                       "grid": list(n), "mixing_rule": args.mixing, "mode": args.mode, "parallelism": "1 GPU",
                       "timed_call": "fg_iterate (no stop-rule hand-over to the host inside the timed region)",
                       "run_load_case_it_s": res.get("run_load_case_it_s")},
            "roofline": roof,
            "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"], "repeats": res["repeats"],
            "sustained_it_s": res.get("sustained_it_s"), "run_load_case_it_s": res.get("run_load_case_it_s"),
            "pcie_inclusive": res.get("pcie_inclusive"),
            # bytes the loop moves as it runs (sum of its kernels' algorithmic bytes) per second of step time -- a rate;
            # loop_GBps_Amin prices the same steps at SURVEY 8d's A_min = 392 B/voxel (strain as the state) so that rounds
            # compare -- an as-if figure, not a rate
            "loop_alg_GBps": res["loop_alg_GBps"], "loop_GBps_Amin": per_voxel * N * it_s / 1e9,
            "kernel_sum_ms": res["kernel_sum_ms"], "hip_event_bias_ms_subtracted": res["hip_event_bias_ms_subtracted"],
            "kernels": kern, "rve": res["rve"],
        }
        out["kernel_sum_note"] = ("sum of the per-kernel HIP-event brackets (event-pair bias subtracted): each bracket includes the "
                                  "kernel's launch gap, so the sum may exceed ms_per_step by a few per cent")
        if not scalar and not stokes:
            # north_star names the Green-operator apply on its own (">= 50 % of the HBM roofline in the Gamma0-apply
            # kernel"): in the default pipeline it is fused into the x pass, so time the stand-alone kernel of the
            # one-kernel-per-routine pipeline as well (96 B per frequency)
            s.set_options(fuse_x=0)
            k2, _, _ = kernel_table(s, E, n, 5, scalar, stokes)
            s.set_options(fuse_x=1)
            if "g0" in k2:
                out["gamma0_apply_standalone"] = {"kernel": "k_g0 (fuse_x=0)", **k2["g0"],
                                                  "frac_of_hbm_peak": k2["g0"]["GBps"] / HBM_PEAK_GBS}
                if 3 * 16 * args.n * args.n * (args.n // 2 + 1) < 2 * 256e6:
                    # the three spectrum components are less than twice the 256 MB Infinity Cache: part of the kernel's bytes
                    # are cache hits (FETCH_SIZE counts them too), a fraction of HBM peak above the copy rate is not an HBM
                    # figure -- the one that is: gamma0_apply_standalone_512 (from the 512^3 leg under `also`)
                    out["gamma0_apply_standalone"]["cache_assisted"] = True
        try:   # the measured roofline: streaming copy / triad of the library on this GPU, 1 GiB arrays
            import ctypes
            from fibergen_amd import _lib
            cg, tg = ctypes.c_double(0.0), ctypes.c_double(0.0)
            if _lib.load().fg_hbm_stream(local_rank, 1024, 3, ctypes.byref(cg), ctypes.byref(tg)) == 0 and tg.value > 0:
                best = max(cg.value, tg.value)
                out["hbm_stream"] = {"copy_GBps": cg.value, "triad_GBps": tg.value,
                                     "dominant_kernel_frac_of_measured": kern[dom]["GBps"] / best}
                if "gamma0_apply_standalone" in out:
                    out["hbm_stream"]["gamma0_apply_frac_of_measured"] = out["gamma0_apply_standalone"]["GBps"] / best
        except Exception as e:  # noqa: BLE001
            out["hbm_stream"] = {"error": "%s: %s" % (type(e).__name__, e)}
        res["cg"] = cg_leg(args, s, E)   # last use of the solver
        if args.method == "cg":
            out.update({"value": res["cg"]["it_s"], "ms_per_step": 1e3 / res["cg"]["it_s"], "cg": res["cg"],
                        "basic_scheme_it_s": it_s})
        else:
            # the reference's default method on the same workload (one operator application + the fused vector sweeps per iteration)
            out["cg_method"] = dict(res["cg"], unit="CG it/s", note="runCGElasticity through fg_run_load_case, maxiter = steps")
        s.close()
        if not args.no_live_traffic and default_options:
            # the counter passes of the dominant kernel on THIS box, now that the solver is closed (children of this process)
            live, info = live_traffic(args, dom)
            if live is not None:
                roof.update({"traffic": live, "traffic_measured_in_this_run": True, "traffic_detail": info,
                             "traffic_over_algorithmic": live / roof["alg_bytes_per_launch"]})
                roof.pop("traffic_from_committed_profile", None)
            else:
                roof["traffic_live_pass_failed"] = info
        out["cache_stream"] = cache_stream(local_rank)
        if args.slab_members > 0 and not scalar and not stokes:
            # the same problem through the slab driver on this one GPU (P = 1: a lone slab, halo = own planes, the
            # all-to-all is the identity; P > 1: all slabs in this process, exchanges are device copies)
            try:
                from fibergen_amd.distributed import SlabGroup
                g = SlabGroup(*n, nranks=args.slab_members, device=local_rank)
                configure(g, phi, normals, args.mixing, args.mode)
                g.calc_ref_material()
                dts = timed_regions(lambda k: g.iterate(E, k), g.synchronize, args.steps, args.warmup, args.repeats)
                med = statistics.median(dts)
                if args.method == "cg":
                    med_cg, _, _ = cg_rate(g, E, args.steps, 5, g.synchronize)
                    out["slab_forced_cg"] = {"members": args.slab_members, "it_s": med_cg,
                                             "ratio_to_single_gpu_cg": med_cg / res["cg"]["it_s"]}
                out["slab_forced"] = {"members": args.slab_members, "it_s": args.steps / med, "ms_per_step": 1e3 * med / args.steps,
                                      "ratio_to_single_gpu_loop": (args.steps / med) / it_s,
                                      "transport": g.members[0].transport,
                                      "note": "the N > 1 path of this script with one slab: what `--gpus N` runs per rank, "
                                              "against the single-GPU loop that is `value`"}
                g.close()
            except Exception as e:  # noqa: BLE001
                out["slab_forced"] = {"error": "%s: %s" % (type(e).__name__, e)}
        also = {}
        cpu_others = []
        for item in [a for a in args.also.split(",") if a]:
            parts = item.split(":")
            ne, mix, mode2 = int(parts[0]), parts[1], (parts[2] if len(parts) > 2 else "elasticity")
            if (ne, mix, mode2) == (args.n, args.mixing, args.mode) or scalar or stokes:
                continue
            key = "%d^3 %s" % (ne, mix) + ("" if mode2 == "elasticity" else " " + mode2)
            try:
                E2 = E
                if mode2 in ("porous", "heat"):
                    E2 = np.array([1.0, 0, 0])
                elif mode2 == "viscosity":
                    E2 = np.array([1.0, -1, 0, 0, 0, 0])
                s2, r2, phi2, nrm2 = measure_single(args, ne, mix, mode2, local_rank, E2, detail=False)
                if mode2 == "elasticity" and 3 * 16 * ne * ne * (ne // 2 + 1) >= 2 * 256e6 and "gamma0_apply_standalone_512" not in out:
                    # the stand-alone Green-operator kernel on a grid the Infinity Cache cannot help with (north_star's
                    # ">= 50 % of the HBM roofline in the Gamma0-apply kernel", SURVEY 8d: 96 B per frequency)
                    s2.set_options(fuse_x=0)
                    k3, _, _ = kernel_table(s2, E2, (ne,) * 3, 5, False, False)
                    s2.set_options(fuse_x=1)
                    if "g0" in k3:
                        out["gamma0_apply_standalone_512" if ne == 512 else "gamma0_apply_standalone_%d" % ne] = {
                            "kernel": "k_g0 (fuse_x=0)", "grid": [ne] * 3, **k3["g0"], "frac_of_hbm_peak": k3["g0"]["GBps"] / HBM_PEAK_GBS}
                s2.close()
                r2["kernels"] = {k: {"avg_ms": v["avg_ms"], "GBps": v["GBps"]} for k, v in r2["kernels"].items()}
                if 3 * 8 * (ne ** 2) * (ne + 2) < 200e6 and out.get("cache_stream"):
                    # three components fit in the Infinity Cache: the fields never leave it between the kernels
                    r2["dominant_kernel"]["frac_of_cache_stream_copy"] = r2["dominant_kernel"]["GBps"] / out["cache_stream"]["copy_GBps"]
                    r2["dominant_kernel"]["note"] = "cache resident: frac_of_hbm_peak is not the yardstick here"
                also[key] = r2
                if mode2 == "elasticity":
                    cpu_others.append((ne, mix, phi2, nrm2))
            except Exception as e:  # noqa: BLE001
                also[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        if also:
            out["also"] = also
        if not args.no_cpu_baseline and not scalar and not stokes:
            try:
                out["cpu_baseline"] = cpu_baseline(n, args.mixing, phi, normals, args.cpu_budget, cpu_others)
            except Exception as e:  # noqa: BLE001  (the GPU measurements above must not be lost with the CPU leg)
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e), "value": None, "unit": "it/s", "kind": "port"}
        if "value" in out.get("cpu_baseline", {}) and out["cpu_baseline"]["value"]:
            # (a ratio against a baseline that stops scaling says little: it is a top-level key only when the sweep grew with
            # the thread count; otherwise it stays inside cpu_baseline beside the note that says so)
            if out["cpu_baseline"]["scales_with_threads"]:
                out["gpu_over_cpu"] = it_s / out["cpu_baseline"]["value"]
            else:
                out["cpu_baseline"]["gpu_over_cpu_at_best_count"] = it_s / out["cpu_baseline"]["value"]
            # north_star's table: it/s per grid on the GPU beside the CPU path timed in this same run
            tab = {"%d^3 %s" % (args.n, args.mixing): {"gpu_it_s": it_s}}
            for k, v in also.items():
                if "it_s" in v:
                    tab[k] = {"gpu_it_s": v["it_s"]}
            for k, v in out["cpu_baseline"]["per_grid"].items():
                if k in tab and "it_s" in v:
                    tab[k].update({"cpu_it_s": v["it_s"], "cpu_threads": v["threads"], "cpu_fft_share": v["fft_share"],
                                   "gpu_over_cpu": tab[k]["gpu_it_s"] / v["it_s"]})
            out["per_grid"] = tab
        print(json.dumps(out), flush=True)
        return

    # ---------------------------------------------------------------------- N > 1: one rank per GPU
    # torch is plumbing for rendez-vous, barrier and the max over ranks.  It must be imported BEFORE libfibergen_amd.so
    # is loaded so that both share one HIP runtime (same soname).
    import torch
    import torch.distributed as dist
    dry = args.dist_backend != "nccl"
    if args.dist_backend == "gloo":
        local_rank = 0
        dist.init_process_group("gloo")
    elif args.dist_backend == "nccl-one-gpu":
        local_rank = 0
        os.environ["NCCL_HOSTID"] = "fibergen-bench-rank-%d" % rank
        os.environ["NCCL_SOCKET_IFNAME"] = "lo"
        os.environ["NCCL_IB_DISABLE"] = "1"
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        if torch.cuda.device_count() < world:   # (counting devices does not initialise the GPU)
            if rank == 0:
                print(json.dumps({"metric": metric, "value": None, "unit": "it/s", "n_gpus": world,
                                  "error": "%d ranks but %d GPUs visible; --dist-backend nccl-one-gpu runs the RCCL path with all "
                                           "ranks on one GPU" % (world, torch.cuda.device_count())}), flush=True)
            sys.exit(5)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # a host-side group for the last barrier: rank 0 times the CPU path there while the others wait in a socket read (a barrier
    # of the nccl group would keep one core per waiting rank spinning beside the CPU measurement)
    try:
        host_group = dist.new_group(backend="gloo") if args.dist_backend != "gloo" else None
    except Exception:  # noqa: BLE001
        host_group = None
    from fibergen_amd import LSSolver
    from fibergen_amd.distributed import DistributedLSSolver
    from fibergen_amd.rve import bench_rve
    if (scalar or stokes) and args.mixing != "voigt":
        raise SystemExit("the scalar and viscosity modes take Voigt mixing")
    phi, normals, par = bench_rve(args.n, args.mixing)
    E_mode = np.array([1.0, 0, 0]) if scalar else (np.array([1.0, -1, 0, 0, 0, 0]) if stokes else None)

    def sync_all(obj):
        obj.synchronize()
        torch.cuda.synchronize(local_rank)
        dist.barrier()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.dist_backend == "gloo" else "cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def pci_bus_id(dev):
        import ctypes
        from fibergen_amd import _lib
        buf = ctypes.create_string_buffer(64)
        return buf.value.decode() if _lib.load().fg_device_pci_bus_id(int(dev), buf, 64) == 0 else "?"

    def exchange_times(solver, passes):
        """per pass, max over ranks: ms of the exchanges measured by HIP events on the exchange stream while stage timing
        was on (kernel_table): every exchange waited for, nothing overlapped -- what the links (and the slowest peer) cost"""
        ct = solver.comm_times()
        return {k: max_over_ranks(v / max(passes, 1)) for k, v in ct.items()}

    devices = [None] * world
    dist.all_gather_object(devices, "%s rank %d: %s" % (socket.gethostname(), rank, pci_bus_id(local_rank)))

    def tune_exchange_mode(solver, E):
        """Which exchange mode the links of THIS node like better is a measurement, not a guess: a few passes with the three
        components in one all-to-all (one message per peer; latency-friendly) and with one all-to-all per component
        overlapping the next component's transforms (bandwidth-friendly), outside the timed region -- like an FFT plan.
        Every rank sees the same (max over ranks) timings and takes the same decision."""
        trials = {}
        for split in (0, 1):
            solver.set_options(slab_split=split)
            solver.iterate(E, 2)
            sync_all(solver)
            t0 = time.perf_counter()
            solver.iterate(E, 4)
            sync_all(solver)
            trials[split] = max_over_ranks((time.perf_counter() - t0) / 4)
        best = min(trials, key=trials.get)
        solver.set_options(slab_split=best)
        return best, {"one_exchange_for_three_components_ms": 1e3 * trials[0], "exchange_per_component_ms": 1e3 * trials[1]}

    replicas = None
    if not args.no_replicas:
        # every rank its own load case (calc_effective_properties' unit strains): no data-path collective
        s = LSSolver(*n, device=local_rank)
        configure(s, phi, normals, args.mixing, args.mode)
        s.calc_ref_material()
        E = np.zeros(6)
        E[rank % 6] = 1.0
        if E_mode is not None:
            E = E_mode
        dts = timed_regions(lambda k: s.iterate(E, k), lambda: sync_all(s), args.steps, args.warmup, 3)
        own = statistics.median(dts)
        med = max_over_ranks(own)
        replicas = {"value": world * args.steps / med, "unit": "it/s", "ms_per_step": 1e3 * med / args.steps,
                    "scaling": "weak", "parallelism": "one load case per GPU x%d, no collective" % world,
                    "rank0_single_gpu_it_s": args.steps / own}
        s.close()

    line = {"metric": metric, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 two-phase fibre RVE (K=%d non-overlapping capsules, R=%.4f, L=%.3f, vf=%.3f), contrast 10, "
                                   "mixing=%s" % (args.n, par["K"], par["R"], par["L"], float(phi.mean()), args.mixing),
                       "grid": list(n), "mixing_rule": args.mixing, "mode": args.mode}}

    def emit(obj, code):
        if rank == 0:
            print(json.dumps(obj), flush=True)
        sys.stdout.flush()
        os._exit(code)

    def watchdog():
        # a stalled exchange must not cost the whole line: report the replicas with the error, exit non-zero
        fb = dict(line)
        fb.update({"value": None, "ms_per_step": None, "scaling": "strong", "replicas": replicas,
                   "slab": {"error": "slab-decomposed section exceeded %d s" % args.slab_timeout}})
        fb["config"] = dict(line["config"], parallelism="x-slabs x%d (slab section stalled: no value; the load-case replicas "
                                                        "are under `replicas`)" % world)
        emit(fb, 3)
    timer = threading.Timer(args.slab_timeout, watchdog)
    timer.daemon = True
    timer.start()
    try:
        if args.n % world:
            raise RuntimeError("grid not divisible by the number of ranks")
        d = DistributedLSSolver(*n, device=local_rank)   # RCCL communicator from an id broadcast over the group
        configure(d, phi, normals, args.mixing, args.mode, slab=d.slab)
        del phi, normals
        d.calc_ref_material()
        E = np.array([1.0, 0, 0, 0, 0, 0]) if E_mode is None else E_mode
        split, split_trials = tune_exchange_mode(d, E)
        dts = timed_regions(lambda k: d.iterate(E, k), lambda: sync_all(d), args.steps, args.warmup, args.repeats)
        med = max_over_ranks(statistics.median(dts))
        lo, hi = max_over_ranks(min(dts)), max_over_ranks(max(dts))
        it_s = args.steps / med
        local_n = (args.n // world, args.n, args.n)
        kern, _, cnt = kernel_table(d, E, local_n, min(args.steps, 10), scalar, stokes)   # this rank's slab, HIP events
        xt = exchange_times(d, cnt)
        dom = max(kern, key=lambda k: kern[k]["avg_ms"])
        line.update({
            "value": it_s, "ms_per_step": 1e3 * med / args.steps, "scaling": "strong",
            "ms_per_step_min": 1e3 * lo / args.steps, "ms_per_step_max": 1e3 * hi / args.steps, "repeats": args.repeats,
            "roofline": {"kernel": dom + " (rank 0's slab)", "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": kern[dom]["GBps"] / HBM_PEAK_GBS, "traffic": None,
                         "alg_bytes_per_launch": int(kern[dom]["alg_GB"] * 1e9), "avg_launch_ms": kern[dom]["avg_ms"]},
            "loop_GBps_Amin": ((24 + 7 * 16) if scalar else A_MIN_BYTES_PER_VOXEL) * N * it_s / 1e9,
            "kernels": kern, "kernel_sum_ms": kernel_table.last_sum_ms,
            "kernel_sum_note": "rank 0's slab; sum of per-kernel HIP-event brackets (event-pair bias subtracted), exchanges excluded",
            "replicas": replicas,
            # the same problem on ONE GPU of this very node (rank 0's replica above): the line states its own speed-up
            "single_gpu_it_s": replicas["rank0_single_gpu_it_s"] if replicas else None,
            "speedup_over_single_gpu": it_s / replicas["rank0_single_gpu_it_s"] if replicas else None,
            "transport": d.transport, "rccl_ranks": world if d.transport == "rccl" else 0,
            "slab_split": split, "slab_split_trials": split_trials,
            "devices": devices, "distinct_devices": len({x.split(": ")[1] for x in devices}),
            "alltoall_ms": xt["alltoall_fwd"] + xt["alltoall_bwd"], "exchange_ms_per_pass": xt,
            "alltoall_MB_per_gpu_per_pass": 2 * (1 if scalar else 3) * (world - 1) / world * (args.n // world) * args.n * (args.n // 2 + 1) * 16 / 1e6,
        })
        line["config"]["parallelism"] = ("x-slabs x%d: ONE problem, displacement loop per slab, per component one RCCL all-to-all "
                                         "each way between the FFT axes, +-1 halo planes of u, norms all-reduced" % world)
        if dry:
            line["config"]["parallelism"] += (" -- DRY RUN: all ranks on one GPU, " +
                                              ("exchanges staged through the host (gloo)" if args.dist_backend == "gloo"
                                               else "RCCL between them over loop-back sockets"))
        if args.method == "cg":
            med_cg, lo_cg, hi_cg = cg_rate(d, E, args.steps, 3, lambda: sync_all(d))
            med_cg = 1.0 / max_over_ranks(1.0 / med_cg)
            line.update({"metric": metric.replace("LS iterations/sec (basic scheme", "CG iterations/sec (runCGElasticity"),
                         "basic_scheme_it_s": it_s, "value": med_cg, "ms_per_step": 1e3 / med_cg})
        d.close()
        timer.cancel()
        if args.also_slab and not scalar and not stokes:
            # the other BASELINE grids (128^3; 512^3 laminate = the north-star target configuration) through the same driver, one
            # after the other; whatever happens here, the headline line stands
            line["also_slab"] = {}
            for item in [a for a in args.also_slab.split(",") if a]:
                def give_up(item=item):
                    line["also_slab"][item] = {"error": "exceeded %d s" % args.slab_timeout}
                    emit(line, 0)
                timer = threading.Timer(args.slab_timeout, give_up)
                timer.daemon = True
                timer.start()
                try:
                    ne, mix = item.split(":")
                    ne = int(ne)
                    if ne % world:
                        raise RuntimeError("grid not divisible by the number of ranks")
                    # every rank generates its own x-slab of the RVE only (the same values as the full field's)
                    phi2, normals2, par2 = bench_rve(ne, mix, x_range=(rank * (ne // world), (rank + 1) * (ne // world)))
                    d2 = DistributedLSSolver(ne, ne, ne, device=local_rank)
                    configure(d2, phi2, normals2, mix, "elasticity")
                    del phi2, normals2
                    d2.calc_ref_material()
                    split2, split_trials2 = tune_exchange_mode(d2, E)
                    steps2 = max(5, args.steps // 2) if ne >= 512 else args.steps
                    dts2 = timed_regions(lambda k: d2.iterate(E, k), lambda: sync_all(d2), steps2, min(args.warmup, 3), 3)
                    med2 = max_over_ranks(statistics.median(dts2))
                    kern2, _, cnt2 = kernel_table(d2, E, (ne // world, ne, ne), min(steps2, 10), False)
                    xt2 = exchange_times(d2, cnt2)
                    line["also_slab"]["%d^3 %s" % (ne, mix)] = {
                        "it_s": steps2 / med2, "ms_per_step": 1e3 * med2 / steps2, "steps": steps2, "repeats": 3,
                        "rve": {"K": par2["K"], "R": par2["R"], "L": par2["L"]}, "transport": d2.transport,
                        "slab_split": split2, "slab_split_trials": split_trials2,
                        "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in kern2.items()},
                        "alltoall_ms": xt2["alltoall_fwd"] + xt2["alltoall_bwd"], "exchange_ms_per_pass": xt2,
                        "alltoall_MB_per_gpu_per_pass": 2 * 3 * (world - 1) / world * (ne // world) * ne * (ne // 2 + 1) * 16 / 1e6}
                    d2.close()
                except Exception as e:  # noqa: BLE001
                    line["also_slab"][item] = {"error": "%s: %s" % (type(e).__name__, e)}
                timer.cancel()
    except Exception as e:  # noqa: BLE001
        line.update({"value": None, "ms_per_step": None, "scaling": "strong", "replicas": replicas,
                     "slab": {"error": "%s: %s" % (type(e).__name__, e)}})
        line["config"]["parallelism"] = ("x-slabs x%d (slab section failed: no value; the load-case replicas are under "
                                         "`replicas`)" % world)
        timer.cancel()
        emit(line, 4)
    timer.cancel()
    try:
        torch.cuda.synchronize(local_rank)
    except Exception:  # noqa: BLE001
        pass
    if rank == 0 and not args.no_cpu_baseline and not scalar and not stokes:
        # north_star's table for N > 1: the CPU path timed in this same run (rank 0, every solver closed; the other ranks wait
        # at the barrier below) and it/s per grid beside it
        try:
            phi, normals, _ = bench_rve(args.n, args.mixing)
            line["cpu_baseline"] = cpu_baseline(n, args.mixing, phi, normals, args.cpu_budget, ())
            tab = {"%d^3 %s" % (args.n, args.mixing): {"gpus": world, "gpu_it_s": line.get("value"),
                                                       "single_gpu_it_s": line.get("single_gpu_it_s"),
                                                       "cpu_it_s": line["cpu_baseline"]["value"],
                                                       "cpu_threads": line["cpu_baseline"]["cores"]}}
            for k, v in (line.get("also_slab") or {}).items():
                if "it_s" in v:
                    tab[k] = {"gpus": world, "gpu_it_s": v["it_s"]}
            line["per_grid"] = tab
        except Exception as e:  # noqa: BLE001
            line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    try:
        dist.barrier(group=host_group) if host_group is not None else dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    main()
