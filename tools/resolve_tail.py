"""Which frames make k_ti_resolve slow: per-frame phase cycles (s_memtime stamps) against the round counts."""
import sys, importlib, numpy as np, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
CH = 100
H, W = 376, 1241
res = torch.zeros((N, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=dev)
s = pkg.Svo(W, H, max_batch=CH); s.track_reset(pkg.Camera(**pkg.KITTI_00_02))
ts = np.zeros((N, 8), np.int64)
for c0 in range(0, N, CH):
    L, R, T = synth.render_sequence(CH, device=dev, start=c0)
    dL = torch.zeros((CH, H, 1280), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    dL[:, :, :W] = L; dR[:, :, :W] = R
    s.track_batch_dev(dL.data_ptr(), dR.data_ptr(), 1280, CH, res[c0:].data_ptr()); s.sync()
    for f in range(CH):
        s.lib.svo_debug_track_stamps(s.h, f, ts[c0 + f].ctypes.data_as(C.c_void_p))
r = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
act1 = r["reserved"][:, 0] & 0xffff; rounds1 = r["reserved"][:, 0] >> 16
act2 = r["reserved"][:, 1] & 0xffff; rounds2 = r["reserved"][:, 1] >> 16
d = np.diff(ts[:, :5], axis=1)
tot = d.sum(1)
order = np.argsort(-tot)
print("frames %d; total cycles: median %d mean %d p90 %d max %d" % (N, np.median(tot), tot.mean(), np.percentile(tot, 90), tot.max()))
print("phase means (begin, pass1, pass2, end):", d[1:].mean(0).astype(int), " medians:", np.median(d[1:], 0).astype(int))
print("slowest frames: frame total | phases | act1 rounds1 act2 rounds2 dense late")
for f in order[:15]:
    print(f, tot[f], d[f], act1[f], rounds1[f], act2[f], rounds2[f], ts[f, 5], ts[f, 6])
v = slice(1, None)          # frame 0 runs no passes (its stamps are zero)
print("correlation of pass-2 cycles with rounds2: %.3f, with dense rows: %.3f, with act2: %.3f" % (
    np.corrcoef(d[v, 2], rounds2[v])[0, 1], np.corrcoef(d[v, 2], ts[v, 5])[0, 1], np.corrcoef(d[v, 2], act2[v])[0, 1]))
A = np.stack([rounds2[v], ts[v, 5], act2[v], np.ones(N - 1)], 1).astype(float)
coef = np.linalg.lstsq(A, d[v, 2].astype(float), rcond=None)[0]
print("pass-2 cycles ~ %.0f * rounds + %.0f * dense + %.1f * act2 + %.0f" % tuple(coef))
