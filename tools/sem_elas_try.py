"""Experiments: the semantic_elas leg (BASELINE configs[4]) under different sub-batch ramps of the dense front end."""
import sys, os, importlib, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench, torch, svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
for ramp in sys.argv[2:] or ["16,32,64"]:
    if ":" in ramp:
        ramp, pct = ramp.split(":")
        os.environ["SVO_DENSE_CU_PERCENT"] = pct
    os.environ["SVO_DENSE_SUB"] = ramp
    os.environ["SVO_ELAS_DEBUG"] = os.environ.get("DBG", "")
    bench.SEM_ELAS_OPTIONS = {"dense_cu_percent": int(os.environ.get("SVO_DENSE_CU_PERCENT", "100")), "dense_two_launch": int(os.environ.get("TWO", "0"))}
    print("options", bench.SEM_ELAS_OPTIONS, end=" ")
    r = bench.semantic_elas_leg(pkg, cam, dL, dR, dev, rec, n=N)
    print("ramp", ramp, "N", N, "fps %.0f" % r["value"], "identical", r.get("cpu_baseline", {}).get("counters_identical_to_gpu"), flush=True)
