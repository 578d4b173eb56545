"""Experiments: the many-sequence leg in a fresh process, after K throw-away tracker contexts (argv[1]) were created and closed;
repeated argv[2] times in the same process."""
import sys, os, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench, torch, svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dL, dR, T = bench.render_frames(synth, 256, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
ref = torch.zeros((256, rec), dtype=torch.uint8, device=dev)
for k in range(max(K, 1)):
    s = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=256)
    s.track_reset(cam)
    s.track_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, 256, ref.data_ptr())
    s.sync(); s.close()
refn = ref.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
for r in range(reps):
    m = bench.multi_sequence_leg(pkg, cam, dL, dR, fb, rec, dev, refn)
    print("K", K, "rep", r, "fps %.0f" % m["value"], m["sequence0_equals_single_chain"], flush=True)
for pre in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):
    if pre == "modes":
        bench.solver_modes_leg(pkg, cam, dL, dR, fb, rec, dev, 256)
    elif pre == "sharded":
        bench.sharded_run(pkg, cam, dL, dR, 256, 2, [0, 0], rec, reference=ref.cpu().numpy())
    elif pre == "fe":
        f = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=128)
        bench.frontend_leg(pkg, cam, dL, dR, 256, fb, dev, None) if False else None
        f.close()
    m = bench.multi_sequence_leg(pkg, cam, dL, dR, fb, rec, dev, refn)
    print("after", pre, "fps %.0f" % m["value"], flush=True)
