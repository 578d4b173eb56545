import sys, importlib, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench, torch, svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = 1024
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
for B in (128, 197, 256):
    svo = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=256)
    svo.set_option("epnp_exact", 0)
    svo.track_reset(cam)
    res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
    def run(c0, c):
        svo.track_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, bench.PITCH, c, res.data_ptr() + c0 * rec)
    run(0, B); svo.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); te = []
    for c0 in range(B, N, B):
        run(c0, min(B, N - c0)); te.append(time.perf_counter())
    svo.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    chain = bench.tail_chain_from_stamps(svo.debug_track_frames(0, min(B, N - B)))
    print("B", B, "fps", (N - B) / dt, "enqueue done after", te[-1] - t0, "total", dt, "period", chain["frame_period_us"], "hyp", chain["k_tp_hyp_us"], "frame", chain["k_tp_frame_us"], "resolve", chain["k_ti_resolve_us"])
    svo.close()
