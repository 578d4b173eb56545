"""Diagnostics: the headline workload (4541 frames, svo_track_batch_dev in 23 steps of 197) under different option sets,
the sequence rendered once.  usage: python tools/option_sweep.py "" "track_group=8" "fe_cu_percent=25,pose_flag=1" ..."""
import importlib, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import svo_loader, bench
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N, B = 4541, 197
cam = pkg.Camera(**pkg.KITTI_00_02)
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
first = None
for opts in sys.argv[1:] or [""]:
    svo = pkg.Svo(bench.W, bench.H, max_batch=B + 10)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        svo.set_option(k, int(v))
    res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
    best = 0.0
    for rep in range(2):
        svo.track_reset(cam)
        svo.track_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, B + 10, res.data_ptr())      # warm-up step (207 frames)
        svo.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(22):
            off = 207 + s * B
            svo.track_batch_dev(dL.data_ptr() + off * fb, dR.data_ptr() + off * fb, bench.PITCH, B, res.data_ptr() + off * rec)
        svo.sync()
        dt = time.perf_counter() - t0
        best = max(best, 22 * B / dt)
    chain = bench.tail_chain_from_stamps(svo.debug_track_frames(0, B))
    got = res.cpu().numpy().tobytes()
    if first is None:
        first = got
    print("%-40s %8.0f frames/s  identical_to_first=%s  period %.1f us  pose busy %.1f us  resolve %.1f us" % (
        opts or "(defaults)", best, got == first, chain["frame_period_us"]["median"], chain["pose_chain_busy_us"]["median"],
        chain["k_ti_resolve_us"]["median"]), flush=True)
    svo.close()
