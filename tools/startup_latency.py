"""Cold-start latency of the library on a GPU box: import, fg_create (HIP initialisation + code object), phases, first and second run at 64^3.
    python tools/startup_latency.py   ->  create 0.2-0.3 s, first run 11 ms, second run 1 ms (22 iterations)"""
import time, sys, os
t0=time.time()
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
t1=time.time()
from fibergen_amd import LSSolver
t2=time.time()
s=LSSolver(64,64,64)
t3=time.time()
from helpers import sphere_phi, lame, MATRIX, INCLUSION
phi=sphere_phi((64,64,64),0.3)
s.set_num_phases(2); m0,m1=lame(**MATRIX),lame(**INCLUSION)
s.set_phase(0,m0[0],m0[1],1-phi); s.set_phase(1,m1[0],m1[1],phi)
t4=time.time()
s.run(np.array([0.01,0,0,0,0,0.0]))
t5=time.time()
s.run(np.array([0.01,0,0,0,0,0.0]))
t6=time.time()
print("numpy %.2f s, import+dlopen %.2f, create %.2f, phases %.2f, first run %.3f (%d it), second run %.3f" % (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,s.iterations,t6-t5))
