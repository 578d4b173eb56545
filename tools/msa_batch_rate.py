"""svo_msa_batch_dev over 64 distinct synth-kitti pairs resident in HBM, five repetitions: pairs/s per repetition (the figure of
bench.py's `msa` leg, without the rest of the leg)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, svo_loader, bench  # noqa: E402
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
dL, dR, _ = bench.render_frames(synth, 64, dev, synth.BASE_SEED)
ctx = pkg.Svo(bench.W, bench.H, device=0)
d_disp = torch.zeros((64, bench.H, bench.W), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
rates = []
for rep in range(6):
    t0 = time.perf_counter()
    ctx.msa_batch_dev(dL.data_ptr(), dR.data_ptr(), dL.stride(1), bench.W, bench.H, 64, d_disp.data_ptr(), 48); ctx.sync()
    rates.append(64 / (time.perf_counter() - t0))
print("svo_msa_batch_dev, 64 pairs: " + " ".join("%.1f" % r for r in rates) + " pairs/s (first call allocates)")
ctx.close()
