"""Where the time of a 805-MB download goes: page faults of the fresh destination, the host copy team, the link.
    python tools/transfer_probe.py [n=256]"""
import ctypes
import json
import mmap
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    from fibergen_amd import LSSolver, _lib
    lib = _lib.load()
    print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:",
          open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip(), flush=True)
    rng = np.random.default_rng(0)
    eps = rng.standard_normal((6, n, n, n))
    hip = ctypes.CDLL("libamdhip64.so")
    libc = ctypes.CDLL("libc.so.6", use_errno=True)
    for staged in (0, 1):
        s = LSSolver(n, n, n)
        s.set_options(staged_copy=staged)
        s.set_num_phases(2)
        s.set_field("epsilon", eps)
        s.synchronize()
        dp = ctypes.POINTER(ctypes.c_double)

        def get(buf):
            t0 = time.perf_counter()
            rc = lib.fg_get_field(s._h, b"epsilon", buf.ctypes.data_as(dp))
            assert rc == 0
            return 1e3 * (time.perf_counter() - t0)
        out = {"staged_copy": staged}
        warm = np.empty_like(eps)
        get(warm)
        out["prefaulted_ms"] = [get(warm) for _ in range(3)]
        fresh = []
        for _ in range(3):
            b = np.empty_like(eps)
            fresh.append(get(b))
            assert np.array_equal(b, eps)
            del b
        out["fresh_np_empty_ms"] = fresh
        hp = []
        for _ in range(3):
            b = np.empty_like(eps)
            a0 = (b.ctypes.data + 4095) & ~4095
            r = libc.madvise(ctypes.c_void_p(a0), ctypes.c_size_t((b.nbytes - 4096) & ~4095), 14)   # MADV_HUGEPAGE
            hp.append(get(b))
            del b
        out["fresh_madv_hugepage_ms"] = hp
        out["madvise_rc"] = r
        # pinning the destination in place
        b = np.empty_like(eps)
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(ctypes.c_void_p(b.ctypes.data), ctypes.c_size_t(b.nbytes), 0)
        t1 = time.perf_counter()
        hip.hipHostUnregister(ctypes.c_void_p(b.ctypes.data))
        t2 = time.perf_counter()
        out["hipHostRegister_fresh_ms"] = [rc, 1e3 * (t1 - t0), 1e3 * (t2 - t1)]
        t0 = time.perf_counter()
        b[:] = 0.0
        out["numpy_touch_ms"] = 1e3 * (time.perf_counter() - t0)
        s.close()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
