"""Where an LM run spends its cycles (library built with -DPOSE_PROF: the statistics fields carry phase sums instead):
build = the normal equations (one pass over the edges per trial + the initial one), serial = thread 0's LDL^T + exp map,
rest = accept / reject logic + hand-over.  k cycles per iteration.
    make -C stereo-semantic-vo_amd ... ; hipcc ... -DPOSE_PROF -c csrc/svo_pose.hip ; SVO_LIB_PATH=<that library> python tools/lm_phases.py"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, svo_loader, util
pkg = svo_loader.load()
svo = pkg.Svo(640, 240)
for mode in (2, 1):
    svo.set_option("pose_mfma", mode)
    print("pose_mfma = %d" % mode)
    for n in (40, 48, 57, 64, 66, 89, 128, 200, 500):
        Xw, obs, K, Tt = util.pose_problem(7, n=n, outlier_frac=0.1)
        rows = []
        for rep in range(3):
            T, st = svo.pose_opt(Xw, obs, K, Tt)
            rows.append((st.iterations, st.chi2_initial, st.chi2_final, st.lambda_final))
        it = rows[-1][0]
        b, s, d = (np.median([r[k] for r in rows]) / it / 1e3 for k in (1, 2, 3))
        print("  n %3d iterations %d: build %5.2f serial %5.2f rest %5.2f  total %5.2f k cycles per iteration" % (n, it, b, s, d, b + s + d))
svo.close()
