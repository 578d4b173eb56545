"""The sharded leg as the FIRST thing a process does (no context before it), then the single-context reference."""
import sys, os, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench, torch, svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
for rep in range(2):
    r = bench.sharded_run(pkg, cam, dL, dR, N, 2, [0, 0], rec)
    print("sharded (rep %d) %.0f frames/s" % (rep, r["value"]), flush=True)
ref = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
s = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=256)
s.track_reset(cam)
import time
t0 = time.perf_counter()
for c0 in range(0, N, 256):
    s.track_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, bench.PITCH, 256, ref.data_ptr() + c0 * rec)
s.sync(); print("single %.0f frames/s" % (N / (time.perf_counter() - t0))); s.close()
r = bench.sharded_run(pkg, cam, dL, dR, N, 2, [0, 0], rec, reference=ref.cpu().numpy())
print("sharded after single %.0f frames/s identical %s" % (r["value"], r["records_identical_to_single_context"]), flush=True)
