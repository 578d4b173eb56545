"""Times svo_msa_solve against the CPU restatement (oracle) and checks they agree.  GPU box only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import svo_loader, util
pkg = svo_loader.load()
from oracle import binding as ob
ob.build()
s = pkg.Svo(1241, 376)
mk = lambda g: np.ascontiguousarray(np.stack([g, np.roll(g, 1, 1), np.roll(g, 1, 0)], 2))
for (W, H, d) in [(320, 120, 48), (1241, 376, 48)]:
    L, R = util.urban_pair(W, H, 0, 0)
    L3, R3 = mk(L), mk(R)
    s.msa_solve(L3, R3, d, 1)
    t = time.time(); g = s.msa_solve(L3, R3, d, 1); tg = time.time() - t
    s.profile_enable(True); s.profile_reset(); s.msa_solve(L3, R3, d, 1); prof = s.profile(); s.profile_enable(False)
    t = time.time(); r = ob.msa_solve(L3, R3, d, 1); tr = time.time() - t
    print("%dx%d d=%d: product %.3f s, restatement %.3f s, differing pixels %d" % (W, H, d, tg, tr, int((g != r).sum())), flush=True)
    for k, v in prof.items():
        if "msa" in k: print("   %-18s %9.3f ms over %d" % (k, v[0], v[1]))
