"""Where the two chains of the ordered tail wait for each other: s_memrealtime stamps per frame over consecutive
svo_track_batch_dev calls (no profiler attached)."""
import sys, time, importlib, numpy as np, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 197
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 376, 1241
dL = torch.zeros((NC * B, H, 1280), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
for c0 in range(0, NC * B, 64):
    c = min(64, NC * B - c0)
    L, R, T = synth.render_sequence(c, device=dev, start=c0)
    dL[c0:c0 + c, :, :W] = L; dR[c0:c0 + c, :, :W] = R
res = torch.zeros((NC * B, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=dev)
s = pkg.Svo(W, H, max_batch=B); s.track_reset(pkg.Camera(**pkg.KITTI_00_02))
rt = np.zeros((NC, B, 4), np.int64)
img = H * 1280
# calls back to back (as bench.py issues them); the stamps of a call are read after the NEXT call has been enqueued
s.sync(); t0 = time.perf_counter()
for c in range(NC):
    s.track_batch_dev(dL.data_ptr() + c * B * img, dR.data_ptr() + c * B * img, 1280, B, res.data_ptr() + c * B * res.shape[1])
    s.sync()   # (work slots are reused by the next call: read them now)
    for f in range(B):
        s.lib.svo_debug_track_realtime(s.h, f, rt[c, f].ctypes.data_as(C.c_void_p))
t1 = time.perf_counter()
for c in range(1, NC):
    r = rt[c].astype(np.float64) / 100.0   # us
    base = r[0, 0]
    idx_s, idx_e, hyp_s, fr_e = r[:, 0] - base, r[:, 1] - base, r[:, 2] - base, r[:, 3] - base
    pose_busy = fr_e - hyp_s
    pose_gap = hyp_s[1:] - fr_e[:-1]
    stall_idx = np.maximum(0, idx_e[1:] - fr_e[:-1])      # pose chain of frame f waits for the index chain of frame f
    print("call %d: span %.2f ms = %.1f us/frame; pose chain busy %.1f us/frame (median %.1f), gaps %.1f us/frame (median %.1f)" % (
        c, (fr_e[-1] - idx_s[0]) / 1e3, (fr_e[-1] - idx_s[0]) / B, pose_busy.mean(), np.median(pose_busy), pose_gap.mean(), np.median(pose_gap)))
    print("   first k_tp_hyp starts %.0f us after the call's first k_ti_resolve; index chain: busy %.1f us/frame, ahead of the pose chain by (frames) %s" % (
        hyp_s[0], (idx_e - idx_s).mean(), [int(np.searchsorted(idx_e, hyp_s[f]) - f) for f in (0, 20, 50, 100, 150, B - 1)]))
    print("   frame period (pose-kernel end to end): mean %.1f median %.1f us; largest 5: %s" % (np.diff(fr_e).mean(), np.median(np.diff(fr_e)), np.sort(np.diff(fr_e))[-5:].round(0)))
    print("   pose chain waiting for the index chain: %.1f us/frame (frames with a wait > 2 us: %d)" % (stall_idx.mean(), int((stall_idx > 2).sum())))
