"""The N > 1 path on CPU: the library's exchange plan (fg_slab_plan -- the peers, buffers, offsets and counts the slab
driver hands to RCCL) executed over gloo, world_size 2 and 4, on NumPy buffers tagged with global indices: the pencil
transpose must deliver exactly the y-slab of the global spectrum and return it, the halo exchanges exactly the
neighbours' boundary planes (periodic).  The kernels themselves run in the GPU suite (tests/test_gpu_slab.py: all P
members on one GPU; tests/test_gpu_distributed.py: one member per process over the same plan)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(nproc, out, *args, timeout=600, _attempt=0):
    """`nproc` worker processes of tests/dist_worker.py with the environment torch.distributed's env:// rendez-vous reads
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port) -- started directly: the elastic
    launcher of `python -m torch.distributed.run` costs an interpreter start and a torch import of its own per test, and the
    GPU suite launches sixty of these.  Raises CalledProcessError (with the failing rank's stderr) like subprocess.run(check=True)."""
    if os.environ.get("FG_TEST_LAUNCHER") == "torchrun":   # the elastic launcher, for comparison
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
        port = str(29500 + (os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "dist_worker.py"),
               "--out", out, *args]
        subprocess.run(cmd, check=True, env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        return [np.load(out + ".%d.npz" % r) for r in range(nproc)]
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nproc), OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--out", out, *args]
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
             for r in range(nproc)]
    import threading
    errs = [b""] * nproc

    def drain(i):   # (a rank that fills its stderr pipe must not block while the others wait for it in a collective)
        errs[i] = procs[i].stderr.read()
    readers = [threading.Thread(target=drain, args=(i,), daemon=True) for i in range(nproc)]
    for t in readers:
        t.start()
    import time as _time
    deadline = _time.monotonic() + timeout
    failed = None
    # poll ALL ranks: a rank r > 0 that dies while rank 0 blocks in a collective is noticed at once, not after rank 0's timeout
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]   # the others may hang in a collective waiting for it: stop them below
        elif all(c == 0 for c in codes):
            break
        elif _time.monotonic() > deadline:
            failed = -1
        else:
            _time.sleep(0.05)
    if failed is not None:
        __import__("time").sleep(1.0)   # let the other ranks report their own error first
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    for t in readers:
        t.join(timeout=5)
    if failed is not None:
        if failed < 0:
            raise subprocess.TimeoutExpired(cmd, timeout, stderr=b"\n".join(errs))
        blob = b"\n".join(errs)
        if _attempt < 3 and (b"EADDRINUSE" in blob or b"ddress already in use" in blob):
            # the port was free when it was probed and taken before rank 0 bound it (parallel test runs): once more, new port
            return launch(nproc, out, *args, timeout=timeout, _attempt=_attempt + 1)
        raise subprocess.CalledProcessError(procs[failed].returncode, cmd, stderr=blob)
    return [np.load(out + ".%d.npz" % r) for r in range(nproc)]


@pytest.mark.parametrize("nproc,grid", [(2, "8,8,6"), (2, "6,4,64"), (4, "8,12,5")])
def test_exchange_plan_over_gloo(tmp_path, nproc, grid):
    res = launch(nproc, str(tmp_path / "p"), "--backend", "plan", "--grid", grid)
    for r in res:
        assert list(r["errors"]) == []


@pytest.mark.parametrize("nproc", [2, 4])
def test_get_field_gather_in_pieces_over_gloo(tmp_path, nproc):
    """GlobalViewSolver._gather: the slabs travel in bounded pieces (ragged last piece), every rank ends with the global
    field, FG_GATHER_LIMIT_GB is read at call time."""
    res = launch(nproc, str(tmp_path / "g"), "--backend", "gather")
    for r in res:
        assert list(r["errors"]) == []


def test_plan_is_consistent_across_ranks():
    """Every send has exactly one matching receive of the same size on the peer, in the same per-pair order."""
    from fibergen_amd.distributed import slab_plan
    nx, ny, nz, P = 16, 8, 10, 4
    for what in range(5):
        plans = [slab_plan(nx, ny, nz, P, r, what, 1)[0] for r in range(P)]
        for a in range(P):
            for b in range(P):
                if a == b:
                    continue
                sends = [o["count"] for o in plans[a] if o["send"] and o["peer"] == b]
                recvs = [o["count"] for o in plans[b] if not o["send"] and o["peer"] == a]
                assert sends == recvs, (what, a, b)
        assert all(o["peer"] != r for r in range(P) for o in plans[r])


def test_plan_rejects_bad_arguments():
    from fibergen_amd import _lib
    lib = _lib.load()
    assert lib.fg_slab_plan(8, 8, 8, 3, 0, 0, 0, None, 0, None) == -1    # 8 not divisible by 3
    assert lib.fg_slab_plan(8, 8, 8, 2, 2, 0, 0, None, 0, None) == -1    # rank out of range
    assert lib.fg_slab_plan(8, 8, 8, 2, 0, 9, 0, None, 0, None) == -1    # unknown exchange
