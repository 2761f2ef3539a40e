import sys, numpy as np, math, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle.ls_oracle import LSOracle
from fibergen_amd import geometry
from fibergen_amd.fg import _Fiber, _normalize_phi
n=64
fib=[_Fiber("capsule",[.5,.5,.5],[1,0,0],0.0,0.2,2), _Fiber("capsule",[.5,.5,.5],[1,0,0],0.0,0.4,1)]
mats=[(1.0,3.63867684478),(3.0,2.0),(5.0,4.0)]
def run(phi, mixing="voigt", normals=None, tol=1e-10):
    o=LSOracle(n,n,n,mats=mats,phis=[phi[0],phi[1],phi[2]],normals=normals,mixing_rule=mixing,tol=tol)
    t=time.time(); o.run([1,1,1,0,0,0]); 
    return o.mean_stress()[0], o.iterations, time.time()-t
phi,nrm,_=geometry.voxelize(fib,(n,n,n),(1,1,1),(0,0,0),3,0,want_normals=True)
phis=_normalize_phi(phi)
print("smooth voigt", run(phis))
for lv in (0,1,2,3):
    p,_,_=geometry.voxelize(fib,(n,n,n),(1,1,1),(0,0,0),3,0,smooth_levels=lv)
    print("levels",lv, run(_normalize_phi(p))[:2], "vf", _normalize_phi(p)[1].mean(), _normalize_phi(p)[2].mean())
# binary
x=(np.arange(n)+.5)/n-.5
r=np.sqrt(x[:,None,None]**2+x[None,:,None]**2+x[None,None,:]**2)
pb=np.zeros((3,n,n,n)); pb[0]=1; pb[1]=(r<0.4); pb[2]=(r<0.2)
print("binary voigt", run(_normalize_phi(pb)))
print("smooth laminate", run(phis,"laminate",nrm))
print("exact vf", 4/3*math.pi*(0.4**3-0.2**3), 4/3*math.pi*0.2**3, "ours", phis[1].mean(), phis[2].mean())
