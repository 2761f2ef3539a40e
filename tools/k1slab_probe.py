import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from fibergen_amd import LSSolver
from helpers import lame, MATRIX, INCLUSION
for shape in ((32, 256, 256), (64, 512, 512)):
    rng = np.random.default_rng(0)
    phi = (rng.random(shape) < 0.15).astype(float)
    s = LSSolver(*shape)
    s.set_num_phases(2)
    m0, m1 = lame(**MATRIX), lame(**INCLUSION)
    s.set_phase(0, m0[0], m0[1], 1 - phi); s.set_phase(1, m1[0], m1[1], phi)
    s.calc_ref_material()
    E = np.array([1.0, 0, 0, 0, 0, 0])
    s.iterate(E, 5)
    s.enable_stage_timing(True); s.iterate(E, 20); t, c = s.stage_times(); s.enable_stage_timing(False)
    print(os.environ.get("FG_TILE_LX"), shape, {k: round(1e3 * v / c, 1) for k, v in t.items() if v > 0}, "us")
    s.close()
