"""Experiments: host time per svo_track_multi_step_dev call vs the device's, before and after a sharded run in this process."""
import sys, os, importlib, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench, torch, svo_loader
import numpy as np
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
dL, dR, T = bench.render_frames(synth, 1024, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
W, H, PITCH = bench.W, bench.H, bench.PITCH

def multi(tag):
    S, msteps = 64, 48
    ms = pkg.Svo(W, H, device=0, max_kp=500, max_batch=S)
    ms.set_option("multi_pipeline", 1)
    mres = torch.zeros((msteps * S, rec), dtype=torch.uint8, device=dev)
    ms.track_multi_reset(S, cam)
    for t in range(2):
        ms.track_multi_step_dev(dL.data_ptr() + t * fb, dR.data_ptr() + t * fb, PITCH, S, mres.data_ptr() + t * S * rec)
    ms.track_multi_reset(S, cam)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    hs = []
    for t in range(msteps):
        a = time.perf_counter()
        ms.track_multi_step_dev(dL.data_ptr() + t * fb, dR.data_ptr() + t * fb, PITCH, S, mres.data_ptr() + t * S * rec)
        hs.append(time.perf_counter() - a)
    b = time.perf_counter()
    ms.sync()
    mdt = time.perf_counter() - t1
    probe = ms.debug_stream_probe()
    ms.close()
    hs = np.array(hs) * 1e3
    print(tag, "probe", probe, "fps %.0f" % (msteps * S / mdt), "host ms/step median %.3f max %.3f sum %.1f; final sync %.1f ms; total %.1f ms"
          % (np.median(hs), hs.max(), hs.sum(), (time.perf_counter() - b) * 1e3, mdt * 1e3), flush=True)

multi("fresh")
multi("fresh again")
what = sys.argv[1] if len(sys.argv) > 1 else "sharded"
if what == "sharded":
    r = bench.sharded_run(pkg, cam, dL, dR, 1024, 2, [0, 0], rec)
    print("sharded", round(r["value"]))
elif what == "sharded1":
    r = bench.sharded_run(pkg, cam, dL, dR, 1024, 1, [0], rec)
    print("sharded G=1", round(r["value"]))
elif what == "batch":
    s = pkg.Svo(W, H, device=0, max_kp=500, max_batch=256)
    s.track_reset(cam)
    res = torch.zeros((1024, rec), dtype=torch.uint8, device=dev)
    for r_ in range(2):
        for c0 in range(0, 1024, 256):
            s.track_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, PITCH, 256, res.data_ptr() + c0 * rec)
    s.sync(); s.close()
elif what == "alloc":
    xs = [torch.zeros((512, H, PITCH), dtype=torch.uint8, device=dev) for _ in range(4)]
    del xs
multi("after " + what)
multi("after " + what + " again")
torch.cuda.empty_cache()
multi("after empty_cache")
