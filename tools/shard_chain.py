"""Where svo_track_sharded_dev spends its time: s_memrealtime stamps per frame of the tail (as tools/chain_times.py reads them
for svo_track_batch_dev) + host-side wall clock of the call's phases, G contexts on ONE GPU."""
import sys, time, importlib, numpy as np, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 3
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
H, W = 376, 1241
N = NC * B
dL = torch.zeros((N, H, 1280), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
for c0 in range(0, N, 64):
    c = min(64, N - c0)
    L, R, T = synth.render_sequence(c, device=dev, start=c0)
    dL[c0:c0 + c, :, :W] = L; dR[c0:c0 + c, :, :W] = R
rec = pkg.TRACK_DTYPE.itemsize
res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
img = H * 1280
cam = pkg.Camera(**pkg.KITTI_00_02)
# reference: one context
s = pkg.Svo(W, H, max_batch=B); s.track_reset(cam)
s.sync(); t0 = time.perf_counter()
for c in range(NC):
    s.track_batch_dev(dL.data_ptr() + c * B * img, dR.data_ptr() + c * B * img, 1280, B, res.data_ptr() + c * B * rec)
s.sync(); t1 = time.perf_counter()
print("single context, %d calls of %d: %.1f us/frame" % (NC, B, (t1 - t0) / N * 1e6))
s.close()
per = (B + G - 1) // G
ctxs = [pkg.Svo(W, H, max_batch=per) for _ in range(G)]
ctxs[0].track_reset(cam)
print("probe (before first tracker call)", ctxs[0].debug_stream_probe())
Ls = [[dL[c * B + g:(c + 1) * B:G].contiguous() for g in range(G)] for c in range(NC)]
Rs = [[dR[c * B + g:(c + 1) * B:G].contiguous() for g in range(G)] for c in range(NC)]
torch.cuda.synchronize()
rt = np.zeros((NC, B, 4), np.int64)
for c in range(NC):
    ctxs[0].sync(); torch.cuda.synchronize()
    h0 = time.perf_counter()
    pkg.Svo.track_sharded_dev(ctxs, [t.data_ptr() for t in Ls[c]], [t.data_ptr() for t in Rs[c]], 1280, B, res.data_ptr() + c * B * rec)
    h1 = time.perf_counter()
    ctxs[0].sync()
    h2 = time.perf_counter()
    for f in range(B):
        ctxs[0].lib.svo_debug_track_realtime(ctxs[0].h, f, rt[c, f].ctypes.data_as(C.c_void_p))
    r = rt[c].astype(np.float64) / 100.0
    base = r[0, 0]
    idx_s, idx_e, hyp_s, fr_e = r[:, 0] - base, r[:, 1] - base, r[:, 2] - base, r[:, 3] - base
    print("call %d: host enqueue %.1f ms, until done %.1f ms = %.1f us/frame; tail span (first k_ti_resolve start -> last pose end) %.1f ms = %.1f us/frame"
          % (c, (h1 - h0) * 1e3, (h2 - h0) * 1e3, (h2 - h0) / B * 1e6, (fr_e[-1] - idx_s[0]) / 1e3, (fr_e[-1] - idx_s[0]) / B))
    print("   pose chain busy %.1f us/frame, frame period median %.1f; index chain busy %.1f us/frame; pose waits for index %.1f us/frame; index ahead by (frames) %s"
          % ((fr_e - hyp_s).mean(), np.median(np.diff(fr_e)), (idx_e - idx_s).mean(), np.maximum(0, idx_e[1:] - fr_e[:-1]).mean(),
             [int(np.searchsorted(idx_e, hyp_s[f]) - f) for f in (0, 20, 100, B - 1)]))
print("probe", ctxs[0].debug_stream_probe())
# as bench.py's sharded leg issues it: reset, then all calls back to back, one sync at the end
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctxs[0].track_reset(cam)
    t1 = time.perf_counter()
    for c in range(NC):
        pkg.Svo.track_sharded_dev(ctxs, [t.data_ptr() for t in Ls[c]], [t.data_ptr() for t in Rs[c]], 1280, B, res.data_ptr() + c * B * rec)
    t2 = time.perf_counter()
    ctxs[0].sync()
    t3 = time.perf_counter()
    print("back to back: reset %.1f ms, enqueue %.1f ms, drain %.1f ms -> %.1f us/frame" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) / N * 1e6))
    rl = np.zeros((B, 4), np.int64)
    for f in range(B):
        ctxs[0].lib.svo_debug_track_realtime(ctxs[0].h, f, rl[f].ctypes.data_as(C.c_void_p))
    r = rl.astype(np.float64) / 100.0
    idx_s, idx_e, hyp_s, fr_e = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    print("   last call: tail span %.1f us/frame, pose busy %.1f, period median %.1f mean %.1f, pose waits for index %.1f us/frame, index gaps > 100 us: %s"
          % ((fr_e[-1] - idx_s[0]) / B, (fr_e - hyp_s).mean(), np.median(np.diff(fr_e)), np.diff(fr_e).mean(), np.maximum(0, idx_e[1:] - fr_e[:-1]).mean(),
             [(int(f), int(g)) for f, g in zip(np.nonzero(np.diff(idx_s) > 150)[0][:12], np.diff(idx_s)[np.diff(idx_s) > 150][:12])]))
for x in ctxs:
    x.close()
