"""Repro: 64 frames over G fresh contexts in ONE svo_track_sharded_dev call, several times in one process."""
import importlib, sys
sys.path.insert(0, ".")
import numpy as np, torch
import svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
M, PITCH = 64, 1280
L, R, _ = synth.render_sequence(M, device=dev)
H, W = L.shape[1], L.shape[2]
dL = torch.zeros((M, H, PITCH), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
dL[:, :, :W] = L; dR[:, :, :W] = R
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
s = pkg.Svo(W, H, max_batch=M); s.track_reset(cam)
res = torch.zeros((M, rec), dtype=torch.uint8, device=dev)
s.track_batch_dev(dL.data_ptr(), dR.data_ptr(), PITCH, M, res.data_ptr()); s.sync(); s.close()
want = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
for it, G in enumerate([2, 2, 2, 3, 2]):
    ctxs = [pkg.Svo(W, H, max_batch=(M + G - 1) // G) for _ in range(G)]
    ctxs[0].track_reset(cam)
    Ls = [dL[g::G].contiguous() for g in range(G)]; Rs = [dR[g::G].contiguous() for g in range(G)]
    out = torch.zeros((M, rec), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    pkg.Svo.track_sharded_dev(ctxs, [t.data_ptr() for t in Ls], [t.data_ptr() for t in Rs], PITCH, M, out.data_ptr())
    ctxs[0].sync()
    got = out.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
    bad = [k for k in range(M) if got[k].tobytes() != want[k].tobytes()]
    print("iteration", it, "G", G, "differing frames", len(bad), bad[:12], flush=True)
    if bad:
        k = bad[0]
        for f in pkg.TRACK_DTYPE.names:
            if not np.array_equal(got[k][f], want[k][f]):
                print("   frame", k, f, got[k][f] if got[k][f].size < 4 else got[k][f].ravel()[:4], want[k][f] if want[k][f].size < 4 else want[k][f].ravel()[:4])
    import ctypes as C
    rt = np.zeros((M, 4), np.int64)
    for f in range(M):
        ctxs[0].lib.svo_debug_track_realtime(ctxs[0].h, f, rt[f].ctypes.data_as(C.c_void_p))
    r = (rt - rt[0, 0]) / 100.0
    print("   overflowed", ctxs[0].track_overflowed(), "frame: idx_start idx_end hyp_start frame_end (us)")
    for f in list(range(0, 12)) + [31, 32, 33]:
        print("   %2d: %9.1f %9.1f %9.1f %9.1f" % (f, r[f, 0], r[f, 1], r[f, 2], r[f, 3]))
    for c in ctxs:
        c.close()
