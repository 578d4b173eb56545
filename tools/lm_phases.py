"""Where an LM run spends its cycles (library built with -DPOSE_PROF: the statistics fields carry phase sums instead):
build = normal equations, serial = thread 0's LDL^T + exp map, chi = the edges' chi2 of a trial, sum = its ordered sum,
rest = accept / reject logic + iteration hand-over.  k cycles per iteration."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, svo_loader, util
pkg = svo_loader.load()
svo = pkg.Svo(640, 240)
for n in (40, 48, 57, 64, 66, 89, 128, 200, 500):
    Xw, obs, K, Tt = util.pose_problem(7, n=n, outlier_frac=0.1)
    rows = []
    for rep in range(3):
        T, st = svo.pose_opt(Xw, obs, K, Tt)
        rows.append((st.iterations, st.chi2_initial, st.chi2_final, st.lambda_final, st.terminated, st.trials_total))
    it = rows[-1][0]
    b, s, c, sm, d = (np.median([r[k] for r in rows]) / it / 1e3 for k in (1, 2, 3, 4, 5))
    print("n %3d iterations %d: build %5.2f serial %5.2f chi %5.2f sum %5.2f rest %5.2f  total %5.2f k cycles per iteration" % (n, it, b, s, c, sm, d, b + s + c + sm + d))
svo.close()
