import sys, time, importlib, torch
sys.path.insert(0, ".")
import svo_loader
pkg = svo_loader.load(); synth = importlib.import_module("stereo_semantic_vo_amd.synth")
W, H, P = 1241, 376, 1280
dev = torch.device("cuda", 0)
cam = pkg.Camera(**pkg.KITTI_00_02)
B = 128
L, R, _ = synth.render_sequence(B, device=dev)
dL = torch.zeros((B, H, P), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
dL[:, :, :W] = L; dR[:, :, :W] = R
torch.cuda.synchronize()
fb = H * P
for nctx in (1, 2, 4, 8):
    pb = B // nctx
    ctxs = [pkg.Svo(W, H, max_batch=pb) for _ in range(nctx)]
    def step():
        for i, c in enumerate(ctxs):
            c.frontend_batch_dev(dL.data_ptr() + i * pb * fb, dR.data_ptr() + i * pb * fb, P, pb, cam)
    for _ in range(3): step()
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(10): step()
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    print(nctx, "contexts:", round(B * 10 / dt), "pairs/s", round(dt / 10 * 1e3, 3), "ms/step")
    for c in ctxs: c.close()
