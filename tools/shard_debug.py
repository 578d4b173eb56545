"""Diagnostics: svo_track_sharded_dev (two contexts on one GPU) against svo_track_batch_dev, frame by frame."""
import importlib, sys
sys.path.insert(0, ".")
import numpy as np, torch
import svo_loader
import bench
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cam = pkg.Camera(**pkg.KITTI_00_02)
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
s = pkg.Svo(bench.W, bench.H, max_batch=512); s.track_reset(cam)
ref = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
for c0 in range(0, N, 512):
    c = min(512, N - c0)
    s.track_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, bench.PITCH, c, ref.data_ptr() + c0 * rec)
s.sync(); s.close()
refn = ref.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
for chunk in (512, 64, 1024):
    for G in (2, 3):
        leg = bench.sharded_run(pkg, cam, dL, dR, N, G, [0] * G, rec, chunk=chunk, reference=ref.cpu().numpy())
        print("chunk", chunk, "G", G, "identical", leg["records_identical_to_single_context"], "fps", round(leg["value"]), flush=True)
# locate the first difference for chunk 512, G 2
per = 256
ctxs = [pkg.Svo(bench.W, bench.H, max_batch=per) for _ in range(2)]
ctxs[0].track_reset(cam)
res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
for c0 in range(0, N, 512):
    c = min(512, N - c0)
    Ls = [dL[c0 + g:c0 + c:2].contiguous() for g in range(2)]; Rs = [dR[c0 + g:c0 + c:2].contiguous() for g in range(2)]
    torch.cuda.synchronize()
    pkg.Svo.track_sharded_dev(ctxs, [t.data_ptr() for t in Ls], [t.data_ptr() for t in Rs], bench.PITCH, c, res.data_ptr() + c0 * rec)
    ctxs[0].sync()
got = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
bad = [k for k in range(N) if got[k].tobytes() != refn[k].tobytes()]
print("differing frames:", len(bad), bad[:10])
if bad:
    k = bad[0]
    for f in pkg.TRACK_DTYPE.names:
        if not np.array_equal(got[k][f], refn[k][f]):
            print(k, f, got[k][f], refn[k][f])
