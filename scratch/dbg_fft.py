import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from helpers import make_gpu_solver
rng=np.random.default_rng(0)
for grid in [(16,16,16),(32,32,32),(64,64,64),(16,16,64),(64,16,16),(16,64,16),(8,8,16),(8,8,32),(128,128,128)]:
    s=make_gpu_solver(grid)
    f=rng.standard_normal((3,)+grid)
    s.set_field("f",f); s.run_stage("fft_forward"); fh=s.get_field("f_hat")
    ref=np.fft.rfftn(f,axes=(1,2,3))/np.prod(grid)
    e1=np.abs(fh-ref).max()/np.abs(ref).max()
    s.set_field("f_hat",ref); s.run_stage("fft_inverse"); u=s.get_field("f")
    nan=np.isnan(u).sum()
    e2=np.nanmax(np.abs(u-f)) if nan<u.size else -1
    print(grid,"fwd err %.2e"%e1,"inv nan",nan,"of",u.size,"err %.2e"%e2, flush=True)
