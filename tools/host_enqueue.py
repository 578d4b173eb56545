"""Host time of one svo_track_batch_dev call (pure enqueue) against the GPU time of the same call."""
import sys, time, importlib, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 197
H, W = 376, 1241
L, R, T = synth.render_sequence(B, device=dev)
dL = torch.zeros((B, H, 1280), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
dL[:, :, :W] = L; dR[:, :, :W] = R
res = torch.zeros((B, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=dev)
s = pkg.Svo(W, H, max_batch=B); s.track_reset(pkg.Camera(**pkg.KITTI_00_02))
for it in range(4):
    s.sync(); t0 = time.perf_counter()
    s.track_batch_dev(dL.data_ptr(), dR.data_ptr(), 1280, B, res.data_ptr())
    t1 = time.perf_counter(); s.sync(); t2 = time.perf_counter()
    print("call %d: host enqueue %.2f ms (%.1f us/frame), until done %.2f ms (%.1f us/frame)" % (it, (t1 - t0) * 1e3, (t1 - t0) * 1e6 / B, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / B))
