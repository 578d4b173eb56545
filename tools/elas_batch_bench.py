"""Throughput of svo_elas_batch_dev on one MI355X: B device-resident pairs per call.
Usage: python tools/elas_batch_bench.py [--batch B] [--iters N]"""
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
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[8, 32, 64])
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--synth", action="store_true", help="distinct synth-kitti frames (bench.py's elas leg) instead of one replicated pair")
a = ap.parse_args()
svo = svo_loader.load()
L, R = util.urban_pair()
H, W = L.shape
stride = 1280
dev = torch.device("cuda", 0)
out = {"workload": "urban1 crop 1241x376 replicated, ROBOTICS", "unit": "stereo pairs/s"}
for B in a.batch:
    ctx = svo.Svo(W, H)
    dL = torch.zeros((B, H, stride), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    dL[:, :, :W] = torch.from_numpy(L).to(dev); dR[:, :, :W] = torch.from_numpy(R).to(dev)
    if a.synth:
        import importlib
        synth = importlib.import_module("stereo_semantic_vo_amd.synth")
        for c0 in range(0, B, 64):
            c = min(64, B - c0)
            Ls, Rs, _ = synth.render_sequence(c, device=dev, start=c0)
            dL[c0:c0 + c, :, :W] = Ls; dR[c0:c0 + c, :, :W] = Rs
        L = dL[B - 1, :, :W].cpu().numpy(); R = dR[B - 1, :, :W].cpu().numpy()
    D1 = torch.zeros((B, H, W), dtype=torch.float32, device=dev); D2 = torch.zeros_like(D1)
    torch.cuda.synchronize()
    ctx.elas_batch_dev(dL.data_ptr(), dR.data_ptr(), stride, W, H, B, D1.data_ptr(), D2.data_ptr())
    t0 = time.perf_counter()
    for _ in range(a.iters):
        ctx.elas_batch_dev(dL.data_ptr(), dR.data_ptr(), stride, W, H, B, D1.data_ptr(), D2.data_ptr())
    dt = (time.perf_counter() - t0) / a.iters
    out["B=%d" % B] = round(B / dt, 1)
    e1, _ = ctx.elas_process(L, R)
    assert np.array_equal(D1[B - 1].cpu().numpy(), e1)
    ctx.close()
print(json.dumps(out))
