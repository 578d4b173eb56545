"""Profiling target: two svo_msa_solve calls (1241x376, d = 48) and nothing else - run under rocprofv3 by tools/legs_profile.sh."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import svo_loader, util
pkg = svo_loader.load()
s = pkg.Svo(1241, 376)
L, R = util.urban_pair(1241, 376, 0, 0)
mk = lambda g: np.ascontiguousarray(np.stack([g, np.roll(g, 1, 1), np.roll(g, 1, 0)], 2))
for _ in range(2):
    g = s.msa_solve(mk(L), mk(R), 48, 1)
print("msa ok", int((g > 0).sum()))
s.close()
