"""Where the pose chain spends its time: s_memtime stamps (shader cycles, ~2.1-2.4 per ns) of the last frame of a batched call, printed
in thousands of cycles.  With the fused pose launch (the default) the frame part's first column includes its wait for the samples."""
import sys, importlib, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import numpy as np, torch, svo_loader, bench
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = 256
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
s = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=64)
s.track_reset(cam)
res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
rows = []
stages = []
for c0 in range(0, N, 64):
    s.track_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, bench.PITCH, 64, res.data_ptr() + c0 * rec)
    ts = (C.c_int64 * 16)()
    s.lib.svo_debug_track_pose_stamps(s.h, ts)
    t = np.array(list(ts), np.float64) / 1000.0
    r = res[c0 + 63].cpu().numpy().view(pkg.TRACK_DTYPE)[0]
    rows.append((t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[9] - t[8], t[10] - t[9], t[11] - t[10], int(r["n_lm_edges"]), int(r["lm_iterations"])))
    stages.append((t[5] - t[2], t[6] - t[5], t[7] - t[6], t[14] - t[7], t[15] - t[14], t[3] - t[15]))
print("samples: gather, to EPnP start, EPnP, consensus | frame part: gather+wait+rule, LM, record | edges, LM iterations   (k cycles)")
for r in rows:
    print("  %5.1f %5.1f %5.1f %5.1f | %5.1f %5.1f %5.1f | %d %d" % r)
print("sample 0's solve: control points + alphas + MtM, 12x12 Jacobi + null vectors, L + betas, Gauss-Newton, R t + reprojection, pick   (k cycles)")
for r in stages:
    print("  %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f" % r)
s.close()
