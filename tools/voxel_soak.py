"""Seeded random scenes beyond the suite's twelve: product voxeliser against its checker (tests/test_gpu_voxelize.py::
test_random_scenes_equal_checker for seeds [lo, hi)).   python tools/voxel_soak.py [lo=12] [hi=212]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_voxelize as t  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 12
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 212
bad, t0, slowest = 0, time.time(), (0.0, -1)
for seed in range(lo, hi):
    t1 = time.time()
    try:
        t.test_random_scenes_equal_checker(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:200], flush=True)
    dt = time.time() - t1
    if dt > slowest[0]:
        slowest = (dt, seed)
print("voxel soak: seeds %d..%d, %d failures, %.1f s, slowest scene %.1f s (seed %d)" % (lo, hi - 1, bad, time.time() - t0, slowest[0], slowest[1]))
