import sys, importlib, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'.')
import svo_loader
pkg=svo_loader.load()
synth=importlib.import_module("stereo_semantic_vo_amd.synth")
dev=torch.device("cuda",0)
N=48
L,R,T=synth.render_sequence(N,device=dev)
H,W=L.shape[1],L.shape[2]
dL=torch.zeros((N,H,1280),dtype=torch.uint8,device=dev); dR=torch.zeros_like(dL)
dL[:,:,:W]=L; dR[:,:,:W]=R
res=torch.zeros((N,pkg.TRACK_DTYPE.itemsize),dtype=torch.uint8,device=dev)
s=pkg.Svo(W,H,max_batch=N); s.track_reset(pkg.Camera(**pkg.KITTI_00_02))
s.track_batch_dev(dL.data_ptr(),dR.data_ptr(),1280,N,res.data_ptr()); s.sync()
r=res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
print("act1",r["reserved"][:,0]&0xffff)
print("rounds1",r["reserved"][:,0]>>16)
print("act2",r["reserved"][:,1]&0xffff)
print("rounds2",r["reserved"][:,1]>>16)
print("pass2",r["n_match_pass2"])

import ctypes as C
ts=np.zeros((N,8),np.int64)
for f in range(N):
    s.lib.svo_debug_track_stamps(s.h, f, ts[f].ctypes.data_as(C.c_void_p))
d=np.diff(ts[:, :5],axis=1)
print("phase cycles (begin, pass1, pass2, end), frames 8..16:")
print(d[8:16])
print("mean", d[4:].mean(0))


print("dense rows at the end of pass 2:", ts[:,5], "rows resolved after round 1:", ts[:,6])

pts=np.zeros(16,np.int64)
s.lib.svo_debug_track_pose_stamps(s.h, pts.ctypes.data_as(C.c_void_p))
print("k_tp_hyp cycles: gather %d, to EPnP %d, EPnP %d, count %d | k_tp_frame: gather+select %d, LM %d, end %d" % (pts[1]-pts[0], pts[2]-pts[1], pts[3]-pts[2], pts[4]-pts[3], pts[9]-pts[8], pts[10]-pts[9], pts[11]-pts[10]))
print("EPnP stages: setup %d, Jacobi %d (%d sweeps), L/rho %d, betas+R,t %d" % (pts[5]-pts[2], pts[6]-pts[5], pts[12], pts[7]-pts[6], pts[3]-pts[7]))
print("  betas stage: initial %d, Gauss-Newton %d, to last SVD %d, SVD+rest %d" % (pts[13]-pts[7], pts[14]-pts[13], pts[15]-pts[14], pts[3]-pts[15]))
