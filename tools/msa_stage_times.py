"""Where one MSA aggregation tree's host time goes on this machine (SVO_MSA_TREE_DEBUG marks of svo_msa_tree, the child-list
form; the batched solver uses svo_msa_tree_rec, which shares every stage but the last): the synth-kitti frame 0, 1241 x 376,
three repetitions, one tree at a time (no other builder running).  Prints the library's own stderr marks."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SVO_MSA_TREE_DEBUG"] = "1"
import numpy as np, svo_loader  # noqa: E402
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
L, R, _ = synth.render_sequence(1)
g = L[0].numpy()
bgr = np.ascontiguousarray(np.stack([g, g, g], 2))
s = pkg.Svo(1241, 376)
o = s.msa_init(bgr, np.ascontiguousarray(np.stack([R[0].numpy()] * 3, 2)))
for rep in range(3):
    sys.stderr.write("[msa tree] ---- repetition %d\n" % rep); sys.stderr.flush()
    pkg.msa_tree(o["m3L"], o["r_graL"], o["c_graL"])
s.close()
