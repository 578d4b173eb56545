import sys, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import svo_loader, util
pkg=svo_loader.load()
from oracle import binding as orc
s=pkg.Svo(640,240)
rng=np.random.default_rng(3)
K=np.array([718.856,718.856,607.1928,185.2157])
rep=(C.c_double*3).in_dll(orc.lib(),"orc_epnp_last_rep")
for sigma in (0.0,0.5):
  for trial in range(6):
    Xw,obs,Kk,Tt=util.pose_problem(trial,n=60,outlier_frac=0.0,sigma=sigma)
    idx=rng.choice(60,5,replace=False)
    R,t=orc.epnp5(Xw[idx],obs[idx],K); ro=np.array(list(rep))
    Rg,tg,rg=s.debug_epnp5(Xw[idx],obs[idx],K)
    print("sigma",sigma,"dR %.2e dt %.2e"%(np.abs(R-Rg).max(),np.abs(t-tg).max()),"rep oracle",ro.round(6),"gpu",rg.round(6))
