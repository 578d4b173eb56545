"""The bench's elas leg on the synthetic frames (512 pairs per call), alone in a process."""
import sys, os, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench, torch, svo_loader
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = 512
dL, dR, T = bench.render_frames(synth, N, dev, synth.BASE_SEED)
for rep in range(3):
    r = bench.elas_leg(pkg, 0, dL, dR, bench.PITCH, N, iters=2)
    print("elas leg %.0f pairs/s, differing %s" % (r["value"], r.get("pixels_differing_from_reference")), flush=True)
