"""Latency of svo_elas_process (host buffers in, host buffers out) on one MI355X, with the per-stage
profile, next to the reference's compiled libelas on one host core (when oracle/_ref is present).
Usage: python tools/elas_bench.py [--iters N] [--setting 0|1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import svo_loader  # noqa: E402
import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--setting", type=int, default=0)
ap.add_argument("--ref-iters", type=int, default=5)
a = ap.parse_args()
svo = svo_loader.load()
L, R = util.urban_pair()
ctx = svo.Svo(util.KITTI_W, util.KITTI_H)
p = svo.elas_default_params(a.setting)
for _ in range(3):
    D1, D2 = ctx.elas_process(L, R, p)
ctx.profile_enable(True); ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(a.iters):
    ctx.elas_process(L, R, p)
dt = (time.perf_counter() - t0) / a.iters
prof = {k: round(v[0] / a.iters, 4) for k, v in ctx.profile().items()}
out = dict(workload="urban1 crop 1241x376, setting %d" % a.setting, ms_per_pair=round(dt * 1e3, 3),
           pairs_per_s=round(1 / dt, 1), valid_fraction=float((D1 >= 0).mean()), stage_ms=prof)
try:
    from oracle import binding as ob
    if ob.ref_elas_lib() is not None:
        pb = ob.ref_elas_params(bool(a.setting))
        ob.ref_elas(L, R, pb)
        t0 = time.perf_counter()
        for _ in range(a.ref_iters):
            ob.ref_elas(L, R, pb)
        out["reference_cpu_ms_per_pair_1core"] = round((time.perf_counter() - t0) / a.ref_iters * 1e3, 2)
except Exception as e:  # noqa: BLE001
    out["reference_cpu_error"] = str(e)
print(json.dumps(out))
