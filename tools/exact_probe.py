"""Diagnostics: one EPnP sample and one RANSAC problem in the epnp_exact mode, timed (run under `timeout`)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import svo_loader, util
pkg = svo_loader.load()
from oracle import binding as orc
orc.build()
K = np.array([718.856, 718.856, 607.1928, 185.2157])
Xw, obs, _, _ = util.pose_problem(1, n=60, outlier_frac=0.0, sigma=0.5)
svo = pkg.Svo(640, 240, max_batch=1)
svo.set_option("epnp_exact", 1)
print("ctx ok", flush=True)
t0 = time.time()
Rg, tg, rg = svo.debug_epnp5(Xw[:5], obs[:5], K)
print("epnp5 exact", time.time() - t0, rg, flush=True)
R, t = orc.epnp5(Xw[:5], obs[:5], K)
print("diff", np.abs(R - Rg).max(), np.abs(t - tg).max(), flush=True)
t0 = time.time()
T, mask, st = svo.pnp_ransac(Xw, obs, K, np.eye(4))
print("ransac exact", time.time() - t0, st.best_hypothesis, st.n_inliers, st.iterations, flush=True)
Tr, mr, sr = orc.pnp_ransac(Xw, obs, K, np.eye(4))
print("oracle", sr.best_hypothesis, sr.n_inliers, sr.iterations, np.abs(T - Tr).max(), flush=True)
