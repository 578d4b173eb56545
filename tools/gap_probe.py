"""Diagnostics: where the pose chain of the headline run waits.  For every frame of a few calls: the gap in front of its
RANSAC kernel, and how late the index chain (its own frame, the end of its event group) was at that moment."""
import importlib, sys
sys.path.insert(0, ".")
import numpy as np, torch
import svo_loader, bench
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
B, NC = 197, int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = B * NC
cam = pkg.Camera(**pkg.KITTI_00_02)
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
svo = pkg.Svo(bench.W, bench.H, max_batch=B)
res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
for rep in range(2):
    svo.track_reset(cam)
    dbg = []
    for c in range(NC):
        svo.track_batch_dev(dL.data_ptr() + c * B * fb, dR.data_ptr() + c * B * fb, bench.PITCH, B, res.data_ptr() + c * B * rec)
        if rep == 1 and c >= 2:      # reading the stamps synchronises: only the last calls of the second pass are looked at one by one
            dbg.append(svo.debug_track_frames(0, B).copy())
    svo.sync()
rt = np.concatenate([d["rt"] for d in dbg]).astype(np.float64) * 0.01     # us
rounds = np.concatenate([d["rounds"] for d in dbg])
for ci, d in enumerate(dbg):
    r = d["rt"].astype(np.float64) * 0.01
    gap = r[1:, 2] - r[:-1, 3]                    # pose end of f-1 -> RANSAC start of f
    idx_late = r[1:, 1] - r[:-1, 3]               # > 0: frame f's matching finished after the pose chain became free
    print("call %d: period mean %.1f median %.1f | gaps: mean %.1f, >5us: %d frames summing %.0f us | frames whose own index chain was late: %d (sum %.0f us)" % (
        ci + 2, np.diff(r[:, 3]).mean(), np.median(np.diff(r[:, 3])), gap.mean(), int((gap > 5).sum()), gap[gap > 5].sum(),
        int((idx_late > 0).sum()), idx_late[idx_late > 0].sum()))
    big = np.argsort(-gap)[:6]
    for f in sorted(big):
        print("   frame %3d gap %6.1f us  own index late by %7.1f us  resolve of f..f+3: %s  rounds %s" % (
            f + 1, gap[f], idx_late[f], (r[f + 1:f + 5, 1] - r[f + 1:f + 5, 0]).round(0), d["rounds"][f + 1:f + 5, 1]))
