"""Latency of the host-buffer entry points (one call at a time, images handed over in host memory) on
one MI355X: svo_orb_extract, svo_stereo_frame, svo_track_frame, svo_elas_process.
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
print(json.dumps(out))
