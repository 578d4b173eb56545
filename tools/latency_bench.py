"""Latency of the host-buffer entry points (one call at a time, images handed over in host memory) on
one MI355X: svo_orb_extract, svo_stereo_frame, svo_track_frame, svo_elas_process - and of the reference's own call pattern
(main.cpp:159-195: imread, one Tracking::Track per frame) through host/stereo_kitti on a 40-frame synth-kitti sequence written
as PNGs (its own "median / mean tracking time" report, main.cpp:200-208).
Usage: python tools/latency_bench.py [--iters N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import svo_loader  # noqa: E402
import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
svo = svo_loader.load()
L, R = util.urban_pair()
ctx = svo.Svo(util.KITTI_W, util.KITTI_H)
cam = svo.Svo.camera(**svo.KITTI_00_02)


def timeit(fn):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    return round((time.perf_counter() - t0) / a.iters * 1e3, 3)


out = {"image": "urban1 crop 1241x376", "unit": "ms per call"}
out["svo_orb_extract"] = timeit(lambda: ctx.orb_extract(L))
out["svo_stereo_frame"] = timeit(lambda: ctx.stereo_frame(L, R, cam))
ctx.track_reset(cam)
out["svo_track_frame"] = timeit(lambda: ctx.track_frame(L, R))
out["svo_elas_process"] = timeit(lambda: ctx.elas_process(L, R))
ctx.close()
try:
    import importlib, re, subprocess, tempfile
    import numpy as np
    from PIL import Image
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    n = 40
    Ls, Rs, _ = synth.render_sequence(n)
    c2 = svo.Svo(util.KITTI_W, util.KITTI_H)
    c2.track_reset(cam)
    t0 = time.perf_counter()
    for k in range(n):
        c2.track_frame(Ls[k].numpy(), Rs[k].numpy())
    out["svo_track_frame_synth_kitti_40_frames"] = round((time.perf_counter() - t0) / n * 1e3, 3)
    c2.close()
    with tempfile.TemporaryDirectory() as td:
        seq = os.path.join(td, "seq")
        os.makedirs(os.path.join(seq, "image_2")); os.makedirs(os.path.join(seq, "image_3"))
        for k in range(n):
            Image.fromarray(np.stack([Ls[k].numpy()] * 3, -1)).save(os.path.join(seq, "image_2", "%06d.png" % k))
            Image.fromarray(np.stack([Rs[k].numpy()] * 3, -1)).save(os.path.join(seq, "image_3", "%06d.png" % k))
        open(os.path.join(seq, "times.txt"), "w").write("".join("%e\n" % (0.1 * k) for k in range(n)))
        y = os.path.join(td, "s.yaml")
        open(y, "w").write("%YAML:1.0\nCamera.fx: 718.856\nCamera.fy: 718.856\nCamera.cx: 607.1928\nCamera.cy: 185.2157\n"
                           "Camera.width: 1241\nCamera.height: 376\nCamera.bf: 386.1448\n")
        exe = os.path.join(ROOT, "stereo-semantic-vo_amd", "host", "stereo_kitti")
        t0 = time.perf_counter()
        p = subprocess.run([exe, "voc", y, seq], capture_output=True, text=True, cwd=td)
        wall = time.perf_counter() - t0
        rep = {"wall_ms_per_frame_incl_png_decode_and_startup": round(wall / n * 1e3, 3), "frames": n, "returncode": p.returncode}
        for key in ("median tracking time", "mean tracking time"):
            m = re.search(key + r"[^0-9]*([0-9.eE+-]+)", p.stdout)
            if m:
                rep[key.replace(" ", "_") + "_s"] = float(m.group(1))
        out["host_stereo_kitti"] = rep
except Exception as e:  # noqa: BLE001
    out["host_stereo_kitti"] = {"error": repr(e)}
print(json.dumps(out))
