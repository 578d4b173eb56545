"""Semantic gating (SURVEY f-3 / rows a-6, a-8): the 8-point fundamental matrix
(pnpmatch::poseEstimation2D_2D, reference src/pnpmatch.cc:302-337) and the detection-box gates
(src/Tracking.cc:61-66, src/frame.cc:198-203, src/pnpmatch.cc:101-144)."""
import importlib

import numpy as np
import pytest


def two_view(seed, n=60, noise=0.0):
    rng = np.random.default_rng(seed)
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    X = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n), rng.uniform(6, 50, n)], 1)
    a = 0.03
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.1, -0.05, -1.0])
    p1 = (K @ X.T).T
    p1 = p1[:, :2] / p1[:, 2:]                  # "current" view
    X2 = (R @ X.T).T + t
    p2 = (K @ X2.T).T
    p2 = p2[:, :2] / p2[:, 2:]                  # "last" view
    return p1 + rng.normal(0, noise, p1.shape), p2 + rng.normal(0, noise, p2.shape)


def numpy_8point(p1, p2):
    """Independent textbook implementation (SVD based)."""
    def norm(p):
        c = p.mean(0)
        s = np.sqrt(2) / np.linalg.norm(p - c, axis=1).mean()
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
        return (p - c) * s, T
    q1, T1 = norm(p1)
    q2, T2 = norm(p2)
    A = np.stack([q2[:, 0] * q1[:, 0], q2[:, 0] * q1[:, 1], q2[:, 0], q2[:, 1] * q1[:, 0], q2[:, 1] * q1[:, 1],
                  q2[:, 1], q1[:, 0], q1[:, 1], np.ones(len(q1))], 1)
    F = np.linalg.svd(A)[2][-1].reshape(3, 3)
    U, S, Vt = np.linalg.svd(F)
    F = U @ np.diag([S[0], S[1], 0]) @ Vt
    F = T2.T @ F @ T1
    return F / F[2, 2]


@pytest.mark.parametrize("noise", [0.0, 0.3])
def test_oracle_and_product_8point(orc, pkg, noise):
    p1, p2 = two_view(3, 80, noise)
    ref = numpy_8point(p1, p2)
    Fo = orc.fundamental_8point(p1, p2)
    Fp = pkg.Svo.fundamental_8point(p1, p2)          # host-side code of the C-ABI library: no GPU needed
    assert np.allclose(Fo, ref, rtol=1e-6, atol=1e-9)
    assert np.allclose(Fp, ref, rtol=1e-6, atol=1e-9)
    assert abs(np.linalg.det(Fp)) < 1e-12 and Fp[2, 2] == 1.0
    h1 = np.c_[p1, np.ones(len(p1))]; h2 = np.c_[p2, np.ones(len(p2))]
    resid = np.abs(np.einsum("ni,ij,nj->n", h2, Fp, h1))
    assert resid.max() < (1e-8 if noise == 0 else 0.05)


def test_8point_degenerate_inputs(orc, pkg):
    p1, p2 = two_view(4, 7)
    assert not orc.fundamental_8point(p1, p2).any() and not pkg.Svo.fundamental_8point(p1, p2).any()
    assert not pkg.Svo.fundamental_8point(np.zeros((0, 2)), np.zeros((0, 2))).any()


BOX = np.array([[500, 760, 200, 330]], np.int32)      # left right top bottom (main.cpp:82-95 order)
BIG = np.array([[200, 1000, 195, 370], [20, 120, 30, 90]], np.int32)


def boxes_for(k):
    # frame 0 carries the small box (exercises Tracking::init's never-reset flag); later frames a
    # large one, so map points created outside a box get matched INSIDE one (epipolar veto path)
    return BOX if k == 0 else BIG


@pytest.fixture(scope="module")
def gated_oracle_run(orc, pkg):
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    L, R, T = synth.render_sequence(5)
    L, R = L.numpy(), R.numpy()
    trk = orc.Tracker(L.shape[2], L.shape[1], pkg.KITTI_00_02)
    out = []
    for k in range(5):
        res, cur = trk.track(L[k], R[k], boxes_for(k))
        out.append((res, cur, trk.F.copy(), trk.vetoes))
    trk.close()
    return L, R, out


def test_oracle_gating_effects(orc, pkg, gated_oracle_run):
    L, R, out = gated_oracle_run
    free = orc.Tracker(L.shape[2], L.shape[1], pkg.KITTI_00_02)
    base = [free.track(L[k], R[k])[0] for k in range(5)]
    free.close()
    # the never-reset `dynamic` flag of Tracking::init: everything after the first boxed keypoint
    # is skipped in frame 0, so far fewer map points are created than without boxes
    assert out[0][0]["n_new_mappoints"] < base[0]["n_new_mappoints"]
    # frames >= 1: F is estimated (non-zero, F[2,2] = 1)
    for k in range(1, 5):
        F = out[k][2].reshape(3, 3)
        assert F[2, 2] == 1.0 and np.abs(F).sum() > 1.0
    # the creation gate keeps the local map smaller than the ungated run
    assert out[4][0]["n_local_map"] < base[4]["n_local_map"]
    # the epipolar veto of pass 1 fires at least once over the run (src/pnpmatch.cc:115-144)
    assert sum(o[3] for o in out) > 0


@pytest.mark.gpu
def test_gpu_gated_tracker_matches_oracle(pkg, gated_oracle_run):
    L, R, out = gated_oracle_run
    svo = pkg.Svo(L.shape[2], L.shape[1], max_batch=1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    for k in range(5):
        res = svo.track_frame(L[k], R[k], boxes=boxes_for(k))
        cur = svo.debug_track_matches()
        ref, ref_cur, ref_F, ref_vetoes = out[k]
        for f in ("n_kp", "n_stereo", "n_match_pass1", "n_match_pass2", "n_lm_edges",
                  "n_new_mappoints", "n_local_map"):
            assert res[f] == ref[f], (k, f, res[f], ref[f])
        # the RANSAC consensus is compared to a tolerance (EPnP's N = 1 candidate depends on the eigen-solver's arbitrary
        # null-space basis, see tests/test_configs.py)
        assert abs(int(res["n_pnp_inliers"]) - int(ref["n_pnp_inliers"])) <= max(2, 0.1 * int(ref["n_lm_edges"])), k
        assert np.array_equal(cur[:ref["n_kp"]], ref_cur[:ref["n_kp"]]), k
        if k > 0:
            F, nv = svo.debug_track_gate()
            assert np.allclose(F.reshape(9), ref_F, rtol=1e-6, atol=1e-9), k
            assert nv == ref_vetoes, (k, nv, ref_vetoes)
        T, Tr = res["Tcw"].reshape(4, 4), ref["Tcw"].reshape(4, 4)
        assert np.abs(T - Tr).max() < 5e-4, k
    svo.close()
