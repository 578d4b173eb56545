import sys, os, time, importlib
sys.path.insert(0, "/root/repo")
import torch, svo_loader, bench
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = 1536
dL, dR, _ = bench.render_frames(synth, N, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
fb = bench.H * bench.PITCH
for B in (256, 384, 512, 768):
    for ov in (1, 2, 3, 4):
        for flags in (pkg.CREATE_TAIL_ALL_CUS, 0):
            fe = pkg.Svo(bench.W, bench.H, max_kp=500, max_batch=B, flags=flags)
            fe.set_option("frontend_overlap", ov)
            d_n = torch.zeros(B, dtype=torch.int32, device=dev); d_depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)
            def run(s):
                off = (s * B) % (N - B + 1)
                fe.frontend_batch_dev(dL.data_ptr() + off * fb, dR.data_ptr() + off * fb, bench.PITCH, B, cam, d_nL=d_n.data_ptr(), d_depth=d_depth.data_ptr())
            for s in range(3): run(s)
            fe.sync()
            steps = max(4, 6144 // B)
            t0 = time.perf_counter()
            for s in range(3, 3 + steps): run(s)
            fe.sync()
            dt = time.perf_counter() - t0
            fe.close()
            print("B %4d overlap %d flags %d: %.1f k pairs/s" % (B, ov, flags, B * steps / dt / 1e3), flush=True)
