"""Times svo_msa_solve against the CPU restatement (oracle) and checks they agree.  GPU box only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import svo_loader, util
pkg = svo_loader.load()
from oracle import binding as ob
ob.build()
s = pkg.Svo(1241, 376)
mk = lambda g: np.ascontiguousarray(np.stack([g, np.roll(g, 1, 1), np.roll(g, 1, 0)], 2))
for (W, H, d) in [(320, 120, 48), (1241, 376, 48)]:
    L, R = util.urban_pair(W, H, 0, 0)
    L3, R3 = mk(L), mk(R)
    s.msa_solve(L3, R3, d, 1)
    t = time.time(); g = s.msa_solve(L3, R3, d, 1); tg = time.time() - t
    s.profile_enable(True); s.profile_reset(); s.msa_solve(L3, R3, d, 1); prof = s.profile(); s.profile_enable(False)
    t = time.time(); r = ob.msa_solve(L3, R3, d, 1); tr = time.time() - t
    print("%dx%d d=%d: product %.3f s, restatement %.3f s, differing pixels %d" % (W, H, d, tg, tr, int((g != r).sum())), flush=True)
    for k, v in prof.items():
        if "msa" in k: print("   %-18s %9.3f ms over %d" % (k, v[0], v[1]))

# batched tracker with MSA depth (svo_track_batch_dev, depth_source = 2): frames in flight on several streams / threads
import torch
W, H, B = 1241, 376, int(os.environ.get("MSA_B", "16"))
L, R = util.urban_pair(W, H, 0, 0)
dev = torch.device("cuda", 0)
pitch = 1280
dL = torch.zeros((B, H, pitch), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
dL[:, :, :W] = torch.from_numpy(L).to(dev); dR[:, :, :W] = torch.from_numpy(R).to(dev)
res = torch.zeros((B, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
t = pkg.Svo(W, H, max_batch=B)
t.set_option("depth_source", 2)
for rep in range(2):
    t.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    t0 = time.time()
    t.track_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, B, res.data_ptr()); t.sync()
    dt = time.time() - t0
    print("tracker, MSA depth, %d frames per call: %.3f s = %.1f frames/s" % (B, dt, B / dt), flush=True)
out = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
print("   n_stereo of frame 0 / last:", int(out[0]["n_stereo"]), int(out[-1]["n_stereo"]))
t.close()
