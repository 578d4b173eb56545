"""Cross-check against the REAL reference code that builds here: libelas (Thirdparty/libelas),
compiled from /root/reference by oracle/Makefile.ref into oracle/_ref/.  It is the dense stereo
solver the reference vendors (a "next" row); here it anchors the sparse
epipolar stereo of the hot path on an independent implementation: on the real urban1 street pair the
per-keypoint disparities of the oracle must agree with libelas' dense map at the same pixels.
Tolerance: two different stereo algorithms, so statistical (median < 0.75 px, 90 % within 2 px)."""
import numpy as np
import pytest

import util
from oracle import binding as ob

pytestmark = pytest.mark.skipif(ob.ref_elas_lib() is None,
                                reason="oracle/_ref not built (needs /root/reference at build time)")


def test_elas_runs_and_is_deterministic():
    L, R = util.urban_pair()
    D1, D2 = ob.ref_elas(L, R)
    E1, E2 = ob.ref_elas(L, R)
    assert np.array_equal(D1, E1) and np.array_equal(D2, E2)
    assert D1.shape == L.shape and (D1 >= 0).mean() > 0.6
    assert D1.max() < 256 and D1[D1 < 0].max() <= -1      # invalid pixels are negative (-10)


def test_sparse_stereo_agrees_with_reference_dense_elas():
    L, R = util.urban_pair()
    D1, _ = ob.ref_elas(L, R)
    s = ob.stereo_frame(L, R, 386.1448, 718.856)
    kp, uR = s["kpL"], s["uR"]
    x = np.rint(kp["x"]).astype(int); y = np.rint(kp["y"]).astype(int)
    both = (uR >= 0) & (D1[y, x] >= 0)
    assert both.sum() > 200
    diff = np.abs((kp["x"] - uR) - D1[y, x])[both]
    assert np.median(diff) < 0.75
    assert (diff < 2.0).mean() > 0.90


def test_ctmf_reference_builds():
    import ctypes, os
    so = os.path.join(os.path.dirname(ob.__file__), "_ref", "libref_ctmf.so")
    assert hasattr(ctypes.CDLL(so), "ctmf")


def test_committed_elas_vectors_are_what_the_reference_produces():
    """tests/golden/elas_vectors.npz (used by the GPU suite when oracle/_ref is absent) == the compiled
    reference, re-run here."""
    import os
    G = np.load(os.path.join(util.GOLDEN, "elas_vectors.npz"))
    for name in ("small", "tiny"):
        for mb in (0, 1):
            k = "%s_%d_" % (name, mb)
            L, R = util.urban_pair(*[int(v) for v in G[k + "crop"]])
            p = ob.ref_elas_params(bool(mb))
            r = ob.ref_elas_staged(L, R, p, tri1=G[k + "tri1"], tri2=G[k + "tri2"])
            assert np.array_equal(r["support"], G[k + "support"])
            assert np.array_equal(r["D1"], G[k + "D1"]) and np.array_equal(r["D2"], G[k + "D2"])
            D1, D2 = ob.ref_elas(L, R, p)
            for side, D in (("1", D1), ("2", D2)):
                exp = G[k + "D" + side].ravel().copy()
                exp[G[k + "process_diff_idx" + side]] = G[k + "process_diff_val" + side]
                assert np.array_equal(exp, D.ravel())
