"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on
the same inputs.  Integer / index / descriptor outputs must be bit-exact; float64 pose
outputs within the tolerances written below."""
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(util.GOLDEN, "vectors.npz"))
POSE_ATOL_T = 1e-6      # metres  (1e-4 m per frame is far inside the reference's noise)
POSE_ATOL_R = 1e-8      # rotation-matrix entries


@pytest.fixture(scope="module")
def svo_kitti(pkg):
    s = pkg.Svo(util.KITTI_W, util.KITTI_H, max_batch=2)
    yield s
    s.close()


@pytest.fixture(scope="module")
def svo_small(pkg):
    s = pkg.Svo(640, 240, max_batch=2)
    yield s
    s.close()


def same_kp(a, b):
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(a[f].view(np.uint32) if a[f].dtype.kind == "f" else a[f],
                              b[f].view(np.uint32) if b[f].dtype.kind == "f" else b[f]), f


def test_loaded_library_is_the_in_tree_hip_extension(pkg):
    pkg.load_library()
    maps = open("/proc/self/maps").read()
    assert "libsvo_hip.so" in maps


# ---- ORB stages ---------------------------------------------------------------------------
@pytest.mark.parametrize("src", ["urban", "blocky"])
def test_pyramid_bit_exact(svo_kitti, orc, src):
    img = util.urban_pair()[0] if src == "urban" else util.blocky_image(11, util.KITTI_W, util.KITTI_H)
    svo_kitti.orb_extract(img)
    levels = orc.pyramid_levels(orc.build_pyramid(img), util.KITTI_W, util.KITTI_H)
    for l in range(8):
        assert np.array_equal(svo_kitti.debug_pyramid_level(0, l), levels[l]), "level %d" % l


@pytest.mark.parametrize("src", ["urban", "blocky"])
def test_fast_corners_bit_exact(svo_kitti, orc, src):
    img = util.urban_pair()[0] if src == "urban" else util.blocky_image(12, util.KITTI_W, util.KITTI_H)
    svo_kitti.orb_extract(img)
    levels = orc.pyramid_levels(orc.build_pyramid(img), util.KITTI_W, util.KITTI_H)
    total = 0
    for l in range(8):
        ref = orc.fast_corners(levels[l])
        got = svo_kitti.debug_fast_corners(0, l)
        # the GPU list is unordered: compare as sets of (y, x, score)
        ref_s = ref[np.lexsort((ref[:, 0], ref[:, 1]))]
        got_s = got[np.lexsort((got[:, 0], got[:, 1]))]
        assert np.array_equal(ref_s, got_s), "level %d: %d vs %d" % (l, len(ref), len(got))
        total += len(ref)
    assert total > 2000


@pytest.mark.parametrize("src", ["urban", "blocky", "flat", "noise"])
def test_orb_extract_bit_exact(svo_kitti, orc, src):
    if src == "urban":
        img = util.urban_pair()[1]
    elif src == "blocky":
        img = util.blocky_image(13, util.KITTI_W, util.KITTI_H)
    elif src == "flat":                       # no corners at all: n == 0
        img = np.full((util.KITTI_H, util.KITTI_W), 77, np.uint8)
    else:                                     # white noise: corner-saturated, exercises caps/ties
        img = np.random.default_rng(1234).integers(0, 256, (util.KITTI_H, util.KITTI_W), dtype=np.uint8)
    kp, desc = svo_kitti.orb_extract(img)
    rkp, rdesc = orc.orb_extract(img)
    assert len(kp) == len(rkp)
    same_kp(kp, rkp)
    assert np.array_equal(desc, rdesc)


def test_orb_extract_candidate_list_overflow_path(pkg, orc):
    """k_fast keeps the scored pixels of a tile in a capped LDS list and scans the whole score tile when a tile has
    more: with the cap forced to 0 and to 7 every / almost every tile takes that path - same keypoints."""
    img = util.urban_pair()[0]
    rkp, rdesc = orc.orb_extract(img)
    for cap in (0, 7, 2048):
        s = pkg.Svo(util.KITTI_W, util.KITTI_H)
        s.set_option("fast_cand_cap", cap)
        kp, desc = s.orb_extract(img)
        s.close()
        same_kp(kp, rkp)
        assert np.array_equal(desc, rdesc), cap
    s = pkg.Svo(util.KITTI_W, util.KITTI_H)
    with pytest.raises(pkg.SvoError):
        s.set_option("fast_cand_cap", 4096)
    s.close()


def test_orb_extract_small_image_bit_exact(svo_small, orc):
    img = util.blocky_image(21, 640, 240)
    kp, desc = svo_small.orb_extract(img)
    rkp, rdesc = orc.orb_extract(img)
    same_kp(kp, rkp)
    assert np.array_equal(desc, rdesc)
    # idempotence: a second call on the same context returns the same bytes
    kp2, desc2 = svo_small.orb_extract(img)
    same_kp(kp, kp2)
    assert np.array_equal(desc, desc2)


# ---- stereo -----------------------------------------------------------------------------------
def test_stereo_real_pair_bit_exact(svo_kitti, orc, pkg):
    L, R = util.urban_pair()
    cam = pkg.Camera(**pkg.KITTI_00_02)
    g = svo_kitti.stereo_frame(L, R, cam)
    r = orc.stereo_frame(L, R, cam.bf, cam.fx)
    same_kp(g["kpL"], r["kpL"]); same_kp(g["kpR"], r["kpR"])
    assert np.array_equal(g["dL"], r["dL"]) and np.array_equal(g["dR"], r["dR"])
    assert np.array_equal(g["uR"].view(np.uint32), r["uR"].view(np.uint32))
    assert np.array_equal(g["depth"].view(np.uint32), r["depth"].view(np.uint32))
    assert (g["depth"] > 0).sum() > 150


@pytest.mark.parametrize("W,H", [(1242, 375), (1226, 370), (752, 480), (643, 241)])
def test_stereo_other_image_sizes_bit_exact(orc, pkg, W, H):
    """The other KITTI odometry sizes, a EuRoC-sized frame and an odd size: every width / height residue the tile and
    row-staging code (clamped border dwords, partial tiles, level widths) has to handle, stride != width included."""
    L, R = util.shifted_pair(W * 7 + H, W, H, disparity=9)
    cam = pkg.Camera(**pkg.KITTI_00_02)
    s = pkg.Svo(W, H, max_batch=1)
    g = s.stereo_frame(L, R, cam)
    s.close()
    r = orc.stereo_frame(L, R, cam.bf, cam.fx)
    same_kp(g["kpL"], r["kpL"]); same_kp(g["kpR"], r["kpR"])
    assert np.array_equal(g["dL"], r["dL"]) and np.array_equal(g["dR"], r["dR"])
    assert np.array_equal(g["uR"].view(np.uint32), r["uR"].view(np.uint32))
    assert np.array_equal(g["depth"].view(np.uint32), r["depth"].view(np.uint32))
    assert len(g["kpL"]) > 300 and (g["depth"] > 0).sum() > 100


def test_stereo_constant_disparity(svo_small, orc, pkg):
    L, R = util.shifted_pair(5, 640, 240, disparity=12)
    cam = pkg.Camera(**pkg.KITTI_00_02)
    g = svo_small.stereo_frame(L, R, cam)
    r = orc.stereo_frame(L, R, cam.bf, cam.fx)
    assert np.array_equal(g["uR"].view(np.uint32), r["uR"].view(np.uint32))
    assert np.array_equal(g["depth"].view(np.uint32), r["depth"].view(np.uint32))
    v = g["depth"] > 0
    assert v.sum() > 100 and np.abs(np.median(g["kpL"]["x"][v] - g["uR"][v]) - 12) < 0.1


def test_disp2depth_and_unproject_golden(svo_small, pkg):
    assert np.array_equal(svo_small.disp2depth(G["d2d_disp"], float(G["d2d_bf"][0])), G["d2d_depth"])
    c = G["un_cam"]
    cam = pkg.Camera(c[0], c[1], c[2], c[3], 386.1448)
    out = svo_small.unproject(G["un_uvz"], cam, G["un_R"], G["un_t"])
    assert np.array_equal(np.isnan(out), np.isnan(G["un_xyz"]))
    ok = ~np.isnan(out)
    assert np.allclose(out[ok], G["un_xyz"][ok], rtol=2e-7, atol=0)


def test_disp2depth_full_kitti_map(svo_kitti, orc):
    rng = np.random.default_rng(5)
    disp = rng.integers(-1, 49, (util.KITTI_H, util.KITTI_W)).astype(np.float32)
    assert np.array_equal(svo_kitti.disp2depth(disp, 386.1448), orc.disp2depth(disp, 386.1448))


# ---- Hamming matching ----------------------------------------------------------------------------
def test_hamming_golden(svo_small):
    assert np.array_equal(svo_small.descriptor_distance(G["ham_a"], G["ham_b"]), G["ham_d"])


def test_argmin_golden_and_oracle(svo_small, orc):
    bi, b, s = svo_small.hamming_argmin(G["m_q"], G["m_t"], G["m_mask"])
    assert np.array_equal(np.stack([bi, b, s], 1), G["m_argmin"])
    q, t = util.planted_descriptors(31, 1500, 500)
    mask = (np.random.default_rng(3).random(500) < 0.3).astype(np.uint8)
    for m in (None, mask, np.ones(500, np.uint8)):
        got = svo_small.hamming_argmin(q, t, m)
        ref = orc.hamming_argmin(q, t, m)
        for a, r in zip(got, ref):
            assert np.array_equal(a, r)


@pytest.mark.parametrize("name,md,ratio", [("p1_14", 14, 0.0), ("p1_15", 15, 0.0),
                                            ("p2_29", 29, 2.0), ("p2_30", 30, 2.0)])
def test_greedy_golden(svo_small, name, md, ratio):
    bi, b, s, acc, asg = svo_small.match_greedy(G["m_q"], G["m_t"], G["m_mask"], md, ratio)
    assert np.array_equal(np.stack([bi, b, s, acc], 1), G["m_greedy_" + name])
    assert np.array_equal(asg, G["m_assigned_" + name])


@pytest.mark.parametrize("M,N,md,ratio", [(500, 500, 15, 0.0), (1500, 500, 30, 2.0), (64, 7, 30, 2.0),
                                          (3, 500, 15, 0.0), (700, 1000, 30, 2.0)])
def test_greedy_against_oracle(svo_small, orc, M, N, md, ratio):
    q, t = util.planted_descriptors(100 + M + N, M, N)
    rng = np.random.default_rng(M)
    assigned = (rng.random(N) < 0.1).astype(np.uint8)
    skip = (rng.random(M) < 0.2).astype(np.uint8)
    got = svo_small.match_greedy(q, t, assigned, md, ratio, q_skip=skip)
    ref = orc.match_greedy(q, t, assigned, md, ratio, q_skip=skip)
    for a, r in zip(got, ref):
        assert np.array_equal(a, r)
    assert got[3].sum() > 0


def test_greedy_ratio_exactly_two_is_rejected(svo_small):
    bi, b, s, acc, _ = svo_small.match_greedy(G["r_q"], G["r_t"], np.zeros(8, np.uint8), 30, 2.0)
    assert (bi[0], b[0], s[0], acc[0]) == (5, 10, 20, 0)


def test_bf_match_against_oracle(svo_small, orc):
    q, t = util.planted_descriptors(77, 500, 500)
    for a, r in zip(svo_small.bf_match(q, t), orc.bf_match(q, t)):
        assert np.array_equal(a, r)


def test_empty_inputs(svo_small):
    z = np.zeros((0, 32), np.uint8)
    assert len(svo_small.descriptor_distance(z, z)) == 0
    bi, b, s = svo_small.hamming_argmin(z, util.random_descriptors(1, 5))
    assert len(bi) == 0
    bi, b, s = svo_small.hamming_argmin(util.random_descriptors(1, 5), z)
    assert list(bi) == [-1] * 5 and list(b) == [256] * 5 and list(s) == [256] * 5


# ---- pose ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["lm", "lm2"])
def test_pose_opt_golden(svo_small, case):
    T, st = svo_small.pose_opt(G[case + "_Xw"], G[case + "_obs"], G[case + "_K"], G[case + "_T0"])
    chi0, chi1, lam, iters = G[case + "_scalars"]
    assert st.iterations == int(iters) and st.trials_total == len(G[case + "_trace"])
    assert np.isclose(st.chi2_initial, chi0, rtol=1e-10) and np.isclose(st.chi2_final, chi1, rtol=1e-7)
    assert np.abs(T[:3, 3] - G[case + "_T"][:3, 3]).max() < POSE_ATOL_T
    assert np.abs(T[:3, :3] - G[case + "_T"][:3, :3]).max() < POSE_ATOL_R


def lm_bits(T, st):
    """everything the LM leaves behind, as bits: the pose and the float64 scalars of its statistics"""
    return (np.ascontiguousarray(T, np.float64).tobytes(), st.n_edges, st.iterations, st.trials_total, st.terminated,
            np.float64(st.chi2_initial).tobytes(), np.float64(st.chi2_final).tobytes(), np.float64(st.lambda_final).tobytes())


@pytest.mark.parametrize("seed,n", [(7, 500), (8, 37), (9, 5), (10, 1), (11, 512), (12, 64), (13, 65), (14, 63), (15, 257), (16, 700)])
def test_pose_opt_against_oracle(svo_small, orc, seed, n):
    """Optimizer::PoseOptimization: BITWISE equal to the CPU restatement - pose, chi2, lambda, every counter (the sums over the
    edges run in insertion order on the matrix core, divisions and square roots are IEEE, the exp map's series is shared).
    Sizes around the 64-edge boundary (wave-local trials), a multiple chunk count, and beyond the 512 edges the LDS holds."""
    Xw, obs, K, T_true = util.pose_problem(seed, n=n)
    for T0 in (np.eye(4), T_true):
        T, st = svo_small.pose_opt(Xw, obs, K, T0)
        Tr, sr, _ = orc.pose_opt(Xw, obs, K, T0)
        assert lm_bits(T, st) == lm_bits(Tr, sr), (seed, n)


def test_pose_opt_zero_edges_keeps_pose(svo_small):
    T0 = np.eye(4); T0[:3, 3] = [1, 2, 3]
    T, st = svo_small.pose_opt(np.zeros((0, 3)), np.zeros((0, 2)), G["lm_K"], T0)
    assert st.n_edges == 0 and np.allclose(T, T0)


# cv::solvePnPRansac (reference src/pnpmatch.cc:227).  The default solver ("epnp_exact" = 2) keeps every IEEE operation of OpenCV's
# loops (oracle/orc_pnp_cv.c) and spreads the independent ones over a wavefront per RANSAC sample: the discrete outcome - which
# sample wins, how many samples the adaptive loop visits, the inlier mask - is IDENTICAL, every sample's R, t and errors are
# bit-identical (tests/test_epnp_ord.py).  The POSE cv::solvePnPRansac hands back went through cv::Rodrigues twice (matrix -> vector in
# the RANSAC callback, vector -> matrix in src/pnpmatch.cc:238) on libm's acos / sin / cos, which the device does not imitate: the
# two double poses agree to ~1e-15, far below the CV_32F the reference stores the pose in (the tracker's float32 poses are
# bit-identical on all 4,541 frames of tests/test_full_length.py).
# (The statistical solver, mode 0, orders its linear algebra differently: tolerances in test_epnp5_candidates_against_oracle.)
PNP_TOL = 1e-9
@pytest.mark.parametrize("seed,n,outliers", [(7, 500, 0.2), (12, 60, 0.2), (21, 200, 0.5), (33, 300, 0.0), (5, 9, 0.0)])
def test_pnp_ransac_against_oracle(svo_small, orc, seed, n, outliers):
    Xw, obs, K, T_true = util.pose_problem(seed, n=n, outlier_frac=outliers)
    T0 = np.eye(4)
    T, mask, st = svo_small.pnp_ransac(Xw, obs, K, T0)
    Tr, mr, sr = orc.pnp_ransac(Xw, obs, K, T0)
    assert (st.ok, st.best_hypothesis, st.n_inliers, st.iterations) == (sr.ok, sr.best_hypothesis, sr.n_inliers, sr.iterations)
    assert st.ok == 1 and np.array_equal(mask, mr)
    assert np.abs(T - Tr).max() < PNP_TOL * (1 + np.abs(Tr).max())
    assert np.array_equal(T.astype(np.float32), Tr.astype(np.float32))      # what the reference keeps of it
    # and it is the right pose: within what 0.5 px of noise on five points allows
    assert np.abs(T[:3, 3] - T_true[:3, 3]).max() < 0.3 and np.abs(T[:3, :3] - T_true[:3, :3]).max() < 5e-3


@pytest.fixture()
def svo0(pkg):
    """A context in mode 0: the statistical wave EPnP (svo_epnp_dev.h), an option since round 4."""
    s = pkg.Svo(640, 240, max_batch=1)
    s.set_option("epnp_exact", 0)
    yield s
    s.close()


def test_epnp5_candidates_against_oracle(svo0, orc):
    """One five-point sample, GPU (statistical wave EPnP, mode 0) vs the CPU restatement of OpenCV's epnp.cpp.  All three beta
    candidates are initialised from eigenvectors of the (arbitrary) null-space basis a five-point system has, and
    EPnP's five Gauss-Newton steps can take them to different local minima of the control-point distance constraints,
    so a single sample is only STATISTICALLY reproducible (in OpenCV itself as well): in most samples both sides end in
    the same minimum - reprojection error equal to 1e-6, pose to 1e-5 m - and in the others each side's winner is
    still a sound solution (error within 0.1 px + 50 % of the other's)."""
    import ctypes as C
    rep_o = (C.c_double * 3).in_dll(orc.lib(), "orc_epnp_last_rep")
    K = np.array([718.856, 718.856, 607.1928, 185.2157])
    rng = np.random.default_rng(3)
    tight = total = 0
    for sigma in (0.0, 0.5, 1.5):
        for trial in range(8):
            Xw, obs, _, T_true = util.pose_problem(trial, n=60, outlier_frac=0.0, sigma=sigma)
            idx = rng.choice(60, 5, replace=False)
            R, t = orc.epnp5(Xw[idx], obs[idx], K)
            ro = np.array(list(rep_o))
            Rg, tg, rg = svo0.debug_epnp5(Xw[idx], obs[idx], K)
            total += 1
            lo, hi = min(rg.min(), ro.min()), max(rg.min(), ro.min())
            assert hi < 0.1 + 1.5 * lo, (sigma, trial, ro, rg)
            if abs(rg.min() - ro.min()) < 1e-6 * (1 + ro.min()):
                tight += 1
                assert np.abs(R - Rg).max() < 1e-6 and np.abs(t - tg).max() < 1e-5, (sigma, trial)
            assert abs(np.linalg.det(Rg) - 1) < 1e-9
    assert tight >= 0.6 * total, (tight, total)


def test_epnp5_basis_independent_candidates_agree_on_every_sample(svo0, orc):
    """The N = 2 and N = 3 beta candidates span the null space instead of picking one of its (arbitrary) basis vectors:
    where both sides' Gauss-Newton runs stay in the same basin their reprojection errors agree closely - checked
    candidate by candidate on every sample, not on the minimum only."""
    import ctypes as C
    rep_o = (C.c_double * 3).in_dll(orc.lib(), "orc_epnp_last_rep")
    K = np.array([718.856, 718.856, 607.1928, 185.2157])
    rng = np.random.default_rng(3)
    agree = {1: 0, 2: 0}
    total = 0
    for sigma in (0.0, 0.5, 1.5):
        for trial in range(8):
            Xw, obs, _, _ = util.pose_problem(trial, n=60, outlier_frac=0.0, sigma=sigma)
            idx = rng.choice(60, 5, replace=False)
            orc.epnp5(Xw[idx], obs[idx], K)
            ro = np.array(list(rep_o))
            _, _, rg = svo0.debug_epnp5(Xw[idx], obs[idx], K)
            total += 1
            for c in (1, 2):
                if abs(rg[c] - ro[c]) < 1e-6 * (1 + ro[c]):
                    agree[c] += 1
    print("epnp5 candidates agreeing with the oracle to 1e-6: N=2 %d, N=3 %d of %d samples" % (agree[1], agree[2], total))
    assert agree[1] >= 0.6 * total and agree[2] >= 0.6 * total, (agree, total)


def test_epnp5_exact_mode_equals_oracle_candidate_by_candidate(pkg, orc):
    """svo_set_option("epnp_exact", 1): OpenCV's loops in their own order, one lane per sample - every candidate's
    reprojection error, R and t equal the CPU restatement's to 1e-9 on every sample (what is left is libm's hypot)."""
    import ctypes as C
    rep_o = (C.c_double * 3).in_dll(orc.lib(), "orc_epnp_last_rep")
    K = np.array([718.856, 718.856, 607.1928, 185.2157])
    rng = np.random.default_rng(3)
    svo = pkg.Svo(640, 240, max_batch=1)
    svo.set_option("epnp_exact", 1)
    worst = 0.0
    for sigma in (0.0, 0.5, 1.5):
        for trial in range(8):
            Xw, obs, _, _ = util.pose_problem(trial, n=60, outlier_frac=0.0, sigma=sigma)
            idx = rng.choice(60, 5, replace=False)
            R, t = orc.epnp5(Xw[idx], obs[idx], K)
            ro = np.array(list(rep_o))
            Rg, tg, rg = svo.debug_epnp5(Xw[idx], obs[idx], K)
            assert np.allclose(rg, ro, rtol=1e-9, atol=1e-9), (sigma, trial, rg, ro)
            assert np.abs(R - Rg).max() < 1e-9 and np.abs(t - tg).max() < 1e-9 * (1 + np.abs(t).max()), (sigma, trial)
            worst = max(worst, np.abs(R - Rg).max(), np.abs(t - tg).max())
    svo.close()
    print("epnp_exact worst |delta| =", worst)


@pytest.mark.parametrize("seed,n,outliers", [(7, 500, 0.2), (12, 60, 0.2), (21, 200, 0.5), (33, 300, 0.0), (5, 9, 0.0), (13, 5, 0.0)])
def test_pnp_ransac_exact_mode_equals_oracle(pkg, orc, seed, n, outliers):
    """cv::solvePnPRansac in the bit-comparable mode: the discrete outcome AND the pose (to 1e-9) are the oracle's."""
    Xw, obs, K, T_true = util.pose_problem(seed, n=n, outlier_frac=outliers)
    svo = pkg.Svo(640, 240, max_batch=1)
    svo.set_option("epnp_exact", 1)
    T, mask, st = svo.pnp_ransac(Xw, obs, K, np.eye(4))
    Tr, mr, sr = orc.pnp_ransac(Xw, obs, K, np.eye(4))
    svo.close()
    assert (st.ok, st.best_hypothesis, st.n_inliers, st.iterations) == (sr.ok, sr.best_hypothesis, sr.n_inliers, sr.iterations)
    assert np.array_equal(mask, mr)
    assert np.abs(T - Tr).max() < 1e-9 * (1 + np.abs(Tr).max())


def test_pnp_ransac_degenerate_counts(svo_small, orc):
    """Fewer than five correspondences: OpenCV returns false - the fallback pose comes back, nothing is an inlier;
    exactly five: the one EPnP model, every point an inlier (RANSACPointSetRegistrator::run, count == modelPoints)."""
    Xw, obs, K, T_true = util.pose_problem(13, n=5, outlier_frac=0.0)
    Tf = np.eye(4); Tf[:3, 3] = [1, 2, 3]
    for k in (0, 3, 4):
        T, mask, st = svo_small.pnp_ransac(Xw[:k], obs[:k], K, Tf)
        Tr, mr, sr = orc.pnp_ransac(Xw[:k], obs[:k], K, Tf)
        assert st.ok == sr.ok == 0 and np.array_equal(T, Tf) and np.array_equal(Tr, Tf) and mask.sum() == 0
    T, mask, st = svo_small.pnp_ransac(Xw, obs, K, Tf)
    Tr, mr, sr = orc.pnp_ransac(Xw, obs, K, Tf)
    assert st.ok == sr.ok == 1 and st.n_inliers == sr.n_inliers == 5 and mask.sum() == 5
    assert np.abs(T - Tr).max() < 1e-3


def test_pnp_ransac_needs_no_prior(svo_small, orc):
    """Large inter-frame motion: the true pose is 40 degrees and several metres away from the previous frame's (the
    fallback argument).  cv::solvePnPRansac has useExtrinsicGuess = false - the samples are solved from their five
    points alone - so the pose is found all the same; round 1's Gauss-Newton-from-the-prior samples could not."""
    rng = np.random.default_rng(4)
    n = 150
    cam = (718.856, 718.856, 607.1928, 185.2157)
    u = rng.uniform(40, 1200, n); v = rng.uniform(40, 340, n); z = rng.uniform(5, 60, n)
    Xc = np.stack([(u - cam[2]) * z / cam[0], (v - cam[3]) * z / cam[1], z], 1)
    ang = np.radians(40.0)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([4.0, -0.5, 7.0])
    Xw = ((R.T @ (Xc - t).T).T).astype(np.float32).astype(np.float64)
    obs = (np.stack([u, v], 1) + rng.normal(0, 0.3, (n, 2))).astype(np.float32).astype(np.float64)
    obs[::7] += 40.0                                   # some gross outliers
    K = np.array(cam)
    T, mask, st = svo_small.pnp_ransac(Xw, obs, K, np.eye(4))
    Tr, mr, sr = orc.pnp_ransac(Xw, obs, K, np.eye(4))
    assert st.ok == 1 and (st.best_hypothesis, st.n_inliers, st.iterations) == (sr.best_hypothesis, sr.n_inliers, sr.iterations)
    assert np.array_equal(mask, mr) and mask[::7].sum() == 0 and mask.sum() > 100
    assert np.abs(T[:3, :3] - R).max() < 5e-3 and np.abs(T[:3, 3] - t).max() < 0.2
    assert np.abs(T - Tr).max() < 1e-2


# ---- batched device path, other sizes, ragged outputs ------------------------------------------------
def test_frontend_batch_dev_equals_per_pair(pkg, orc):
    """svo_frontend_batch_dev over HBM-resident, pitched images == the per-pair host API == oracle."""
    import torch
    W, H, B, pitch = 640, 240, 3, 704
    cam = pkg.Camera(**pkg.KITTI_00_02)
    pairs = [util.shifted_pair(30 + i, W, H, disparity=6 + 9 * i) for i in range(B)]
    dev = torch.device("cuda", 0)
    dL = torch.zeros((B, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    for i, (L, R) in enumerate(pairs):
        dL[i, :, :W] = torch.from_numpy(L).to(dev); dR[i, :, :W] = torch.from_numpy(R).to(dev)
    kp = torch.zeros((B, 500, 28), dtype=torch.uint8, device=dev)
    desc = torch.zeros((B, 500, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev)
    uR = torch.zeros((B, 500), dtype=torch.float32, device=dev)
    depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    svo = pkg.Svo(W, H, max_batch=B)
    svo.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, B, cam, kp.data_ptr(), desc.data_ptr(),
                           n.data_ptr(), uR.data_ptr(), depth.data_ptr())
    svo.sync()
    n_h = n.cpu().numpy(); kp_h = kp.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, 500)
    desc_h = desc.cpu().numpy(); uR_h = uR.cpu().numpy(); depth_h = depth.cpu().numpy()
    for i, (L, R) in enumerate(pairs):
        r = orc.stereo_frame(L, R, cam.bf, cam.fx)
        m = len(r["kpL"])
        assert n_h[i] == m
        same_kp(kp_h[i][:m], r["kpL"])
        assert np.array_equal(desc_h[i][:m], r["dL"])
        assert np.array_equal(uR_h[i][:m].view(np.uint32), r["uR"].view(np.uint32))
        assert np.array_equal(depth_h[i][:m].view(np.uint32), r["depth"].view(np.uint32))
    svo.close()


def test_pyramid_one_launch_per_level_path_bit_exact(pkg, orc):
    """The pyramid is built by three fused launches by default (levels kept in LDS); the launch-per-level kernel stays as
    the fallback for images too wide for that - both must give the oracle's levels."""
    img = util.urban_pair()[0]
    levels = orc.pyramid_levels(orc.build_pyramid(img), util.KITTI_W, util.KITTI_H)
    for fused in (0, 1):
        svo = pkg.Svo(util.KITTI_W, util.KITTI_H)
        svo.set_option("pyr_fused", fused)
        svo.orb_extract(img)
        for l in range(8):
            assert np.array_equal(svo.debug_pyramid_level(0, l), levels[l]), "fused %d level %d" % (fused, l)
        svo.close()


def test_frontend_batch_full_size_is_independent_of_scheduling_switches(pkg):
    """KITTI-size batch: keypoints, descriptors and depths from svo_frontend_batch_dev must not depend on how the work is
    scheduled - the pyramid as three fused launches or one per level, one chain or slices on several streams."""
    import torch
    W, H, B, pitch = util.KITTI_W, util.KITTI_H, 16, 1280
    cam = pkg.Camera(**pkg.KITTI_00_02)
    dev = torch.device("cuda", 0)
    L0, R0 = util.urban_pair()
    dL = torch.zeros((B, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    for i in range(B):       # different content per pair: shifted copies of the real crop
        dL[i, :, :W] = torch.from_numpy(np.roll(L0, 7 * i, axis=1)).to(dev)
        dR[i, :, :W] = torch.from_numpy(np.roll(R0, 7 * i, axis=1)).to(dev)
    torch.cuda.synchronize()
    outs = []
    for fused, slices in ((1, 2), (0, 1), (1, 1), (0, 2)):
        kp = torch.zeros((B, 500, 28), dtype=torch.uint8, device=dev)
        desc = torch.zeros((B, 500, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(B, dtype=torch.int32, device=dev)
        depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)
        svo = pkg.Svo(W, H, max_batch=B)
        svo.set_option("pyr_fused", fused)
        svo.set_option("frontend_overlap", slices)
        svo.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, B, cam, kp.data_ptr(), desc.data_ptr(), n.data_ptr(),
                               None, depth.data_ptr())
        svo.sync()
        n_h = n.cpu().numpy(); kp_h = kp.cpu().numpy(); desc_h = desc.cpu().numpy(); depth_h = depth.cpu().numpy()
        assert n_h.min() > 300
        outs.append((n_h.tobytes(),) + tuple(a[i, :n_h[i]].tobytes() for i in range(B) for a in (kp_h, desc_h, depth_h)))
        svo.close()
    for o in outs[1:]:
        assert o == outs[0]


@pytest.mark.parametrize("slices", [1, 2, 3])
def test_frontend_batch_slices_on_streams_equal_oracle(pkg, orc, slices):
    """svo_frontend_batch_dev runs slices of a batch side by side on their own streams (scheduling only): every pair's
    outputs must still be the oracle's, whatever the number of slices."""
    import torch
    W, H, B, pitch = 640, 240, 24, 704
    cam = pkg.Camera(**pkg.KITTI_00_02)
    pairs = [util.shifted_pair(70 + i, W, H, disparity=4 + (5 * i) % 40) for i in range(B)]
    dev = torch.device("cuda", 0)
    dL = torch.zeros((B, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    for i, (L, R) in enumerate(pairs):
        dL[i, :, :W] = torch.from_numpy(L).to(dev); dR[i, :, :W] = torch.from_numpy(R).to(dev)
    kp = torch.zeros((B, 500, 28), dtype=torch.uint8, device=dev)
    desc = torch.zeros((B, 500, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev)
    uR = torch.zeros((B, 500), dtype=torch.float32, device=dev)
    depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    svo = pkg.Svo(W, H, max_batch=B)
    svo.set_option("frontend_overlap", slices)
    for _ in range(2):   # the second call reuses streams / working set slices
        svo.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, B, cam, kp.data_ptr(), desc.data_ptr(),
                               n.data_ptr(), uR.data_ptr(), depth.data_ptr())
    svo.sync()
    n_h = n.cpu().numpy(); kp_h = kp.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, 500)
    desc_h = desc.cpu().numpy(); depth_h = depth.cpu().numpy()
    for i in (0, 7, 8, 11, 12, 15, 16, 23):
        L, R = pairs[i]
        r = orc.stereo_frame(L, R, cam.bf, cam.fx)
        m = len(r["kpL"])
        assert n_h[i] == m, "pair %d" % i
        same_kp(kp_h[i][:m], r["kpL"])
        assert np.array_equal(desc_h[i][:m], r["dL"])
        assert np.array_equal(depth_h[i][:m].view(np.uint32), r["depth"].view(np.uint32))
    svo.close()


@pytest.mark.parametrize("W,H", [(1344, 391), (333, 207), (96, 96), (2048, 96)])
def test_orb_other_image_sizes(pkg, orc, W, H):
    if (W, H) == (1344, 391):
        img = util.read_pgm(os.path.join(util.GOLDEN, "urban1_left.pgm"))
    else:
        img = util.blocky_image(W * 7 + H, W, H)
    svo = pkg.Svo(W, H)
    kp, desc = svo.orb_extract(img)
    rkp, rdesc = orc.orb_extract(img)
    assert len(kp) == len(rkp)
    same_kp(kp, rkp)
    assert np.array_equal(desc, rdesc)
    svo.close()


def test_orb_ragged_fewer_than_500_keypoints(svo_kitti, orc):
    """Texture in one corner only: far fewer than 500 keypoints, several empty pyramid levels."""
    img = np.full((util.KITTI_H, util.KITTI_W), 90, np.uint8)
    img[100:150, 300:370] = util.blocky_image(77, 70, 50)
    kp, desc = svo_kitti.orb_extract(img)
    rkp, rdesc = orc.orb_extract(img)
    assert 0 < len(rkp) < 500 and len(kp) == len(rkp)
    same_kp(kp, rkp)
    assert np.array_equal(desc, rdesc)


def test_invalid_arguments_are_rejected(pkg, svo_small):
    import ctypes as C
    lib = pkg.load_library()
    h = C.c_void_p()
    assert lib.svo_create(C.byref(h), 99, 640, 240, 500, 1) == -2          # no such device
    assert lib.svo_create(C.byref(h), 0, 640, 240, 1000, 1) == -1          # max_kp out of range
    n = C.c_int32(0)
    assert lib.svo_orb_extract(svo_small.h, None, 640, None, None, C.byref(n)) == -1
    with pytest.raises(pkg.SvoError):                                       # capacity: max_batch = 2
        svo_small.frontend_batch_dev(1, 1, 640, 5, pkg.Camera(**pkg.KITTI_00_02))


# ---- MFMA variant of the J^T W J accumulation ------------------------------------------------------
@pytest.mark.parametrize("seed,n", [(7, 500), (8, 37), (9, 5), (10, 1), (11, 512), (17, 130)])
def test_pose_opt_matrix_core_sums_equal_the_one_lane_loop(pkg, orc, seed, n):
    """svo_set_option("pose_mfma"): 1 (default) - the sums over the edges four at a time on v_mfma_f64_4x4x4 (A = 1.0: an in-order
    IEEE sum); 2 - one lane per quantity, plain additions in order; 0 - the whole loop walked by ONE lane, edge by edge (the
    checker).  All three bitwise equal to the CPU restatement."""
    Xw, obs, K, _ = util.pose_problem(seed, n=n, outlier_frac=0.3)
    svo = pkg.Svo(640, 240)
    T0 = np.eye(4)
    Tr, sr, _ = orc.pose_opt(Xw, obs, K, T0)
    for mode in (0, 1, 2):
        svo.set_option("pose_mfma", mode)
        Tm, sm = svo.pose_opt(Xw, obs, K, T0)
        assert lm_bits(Tm, sm) == lm_bits(Tr, sr), mode
    svo.close()


@pytest.mark.parametrize("case", ["lm", "lm2"])
@pytest.mark.parametrize("mode", [1, 2])
def test_pose_opt_mfma_golden(pkg, case, mode):
    svo = pkg.Svo(640, 240)
    svo.set_option("pose_mfma", mode)
    T, st = svo.pose_opt(G[case + "_Xw"], G[case + "_obs"], G[case + "_K"], G[case + "_T0"])
    svo.close()
    chi0, chi1, lam, iters = G[case + "_scalars"]
    assert st.iterations == int(iters) and st.trials_total == len(G[case + "_trace"])
    assert np.isclose(st.chi2_initial, chi0, rtol=1e-10) and np.isclose(st.chi2_final, chi1, rtol=1e-7)
    assert np.abs(T - G[case + "_T"]).max() < POSE_ATOL_T


def test_full_size_batch_properties(pkg):
    """BASELINE-sized batch (128 pairs of 1241x376, what bench.py times), through size-independent properties:
    the batch is deterministic (two runs byte-identical), permuting the pairs permutes the outputs, every pair
    equals the same pair processed alone, and depth = bf / (x - uR) wherever a depth exists."""
    import importlib
    import torch
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    W, H, B, pitch = util.KITTI_W, util.KITTI_H, 128, 1280
    cam = pkg.Camera(**pkg.KITTI_00_02)
    dev = torch.device("cuda", 0)
    L, R, _ = synth.render_sequence(8, device=dev)
    dL = torch.zeros((B, H, pitch), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    for i in range(B):      # 8 rendered frames, repeated with a cyclic column shift so that all 128 pairs differ
        s = 3 * (i // 8)
        dL[i, :, :W] = torch.roll(L[i % 8].to(dev), s, 1); dR[i, :, :W] = torch.roll(R[i % 8].to(dev), s, 1)

    def run(svo, l, r, b):
        kp = torch.zeros((b, 500, 28), dtype=torch.uint8, device=dev)
        desc = torch.zeros((b, 500, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(b, dtype=torch.int32, device=dev)
        uR = torch.zeros((b, 500), dtype=torch.float32, device=dev)
        depth = torch.zeros((b, 500), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        svo.frontend_batch_dev(l.data_ptr(), r.data_ptr(), pitch, b, cam, kp.data_ptr(), desc.data_ptr(), n.data_ptr(),
                               uR.data_ptr(), depth.data_ptr())
        svo.sync()
        return [t.cpu().numpy() for t in (n, kp, desc, uR, depth)]

    svo = pkg.Svo(W, H, max_batch=B)
    a = run(svo, dL, dR, B)
    b = run(svo, dL, dR, B)
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()                       # deterministic
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    c = run(svo, dL[perm.to(dev)].contiguous(), dR[perm.to(dev)].contiguous(), B)
    n = a[0]
    assert n.min() > 300
    for j, i in enumerate(perm.tolist()):                       # permutation equivariance
        m = int(n[i])
        assert c[0][j] == m
        for q in range(1, 5):
            assert a[q][i][:m].tobytes() == c[q][j][:m].tobytes(), (i, q)
    one = pkg.Svo(W, H, max_batch=1)
    for i in (0, 37, 127):                                      # each pair alone == inside the batch
        s = run(one, dL[i:i + 1].contiguous(), dR[i:i + 1].contiguous(), 1)
        m = int(n[i])
        assert s[0][0] == m
        for q in range(1, 5):
            assert a[q][i][:m].tobytes() == s[q][0][:m].tobytes(), (i, q)
    one.close()
    kp = a[1].view(pkg.KP_DTYPE).reshape(B, 500)
    for i in range(B):                                          # depth is bf / disparity, float32, wherever it exists
        m = int(n[i]); d = a[4][i][:m]; has = d > 0
        assert has.sum() > 150
        disp = kp[i]["x"][:m][has] - a[3][i][:m][has]
        assert np.array_equal(d[has], (np.float32(cam.bf) / disp).astype(np.float32))
    svo.close()
