import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import svo_loader, util
pkg=svo_loader.load()
from oracle import binding as orc
s=pkg.Svo(640,240)
K=np.array([718.856,718.856,607.1928,185.2157])
for seed,n in ((1,27),(2,41),(3,200),(4,500)):
    Xw,obs,Kk,Tt=util.pose_problem(seed,n=n)
    outs=[s.pnp_ransac(Xw,obs,K,np.eye(4)) for _ in range(4)]
    Tr,mr,sr=orc.pnp_ransac(Xw,obs,K,np.eye(4))
    same=all(np.array_equal(outs[0][0],o[0]) for o in outs)
    print(n,"deterministic",same,"best",[o[2].best_hypothesis for o in outs],"oracle best",sr.best_hypothesis,"inl",[o[2].n_inliers for o in outs],sr.n_inliers,"iters",outs[0][2].iterations,sr.iterations,"dT",np.abs(outs[0][0]-Tr).max())
