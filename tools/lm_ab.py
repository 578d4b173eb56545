import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import svo_loader, util
pkg = svo_loader.load()
svo = pkg.Svo(640, 240)
for n in (64, 200, 500):
    Xw, obs, K, _ = util.pose_problem(7, n=n)
    for mf in (0, 1):
        svo.set_option("pose_mfma", mf)
        svo.profile_reset(); svo.profile_enable(True)
        for _ in range(50): svo.pose_opt(Xw, obs, K, np.eye(4))
        svo.profile_enable(False)
        ms, cnt = svo.profile()["k_pose_opt"]
        print("n", n, "mfma", mf, "k_pose_opt avg us", round(ms / cnt * 1e3, 1))
