"""BASELINE.json configs that round 1 left untested as a whole.

configs[0]  "KITTI seq 04 via Stereo/KITTI04-12.yaml ... YOLO disabled": the KITTI04-12 intrinsics
            (reference Stereo/KITTI04-12.yaml:8-11,25) through the stereo front end and through a tracked
            sequence, GPU against the oracle (the CPU leg of the same config runs without a GPU).
configs[4]  "semantic mask + libelas dense stereo fused": depth_source = 1 TOGETHER with offline detection boxes
            (reference src/Tracking.cc:225-228 + src/frame.cc:198-203 + src/pnpmatch.cc:101-144), against the
            oracle tracker fed with the maps of the reference's own compiled libelas.
Long run    >= 64 full-size frames GPU vs oracle, frame by frame: culling (from frame 4 on), pool compaction,
            the re-evaluation rounds of the greedy passes, and a no-match sequence that drives the map-point pool to its
            largest reachable size.
"""
import importlib

import numpy as np
import pytest

# The tests below run the device tracker in its DEFAULT mode (svo_set_option "epnp_exact" = 2: every RANSAC sample solved with
# OpenCV's operations and rounding, spread over a wavefront; g2o's LM with its operations, its sums over the edges in insertion
# order) beside a FREE-RUNNING oracle (no teacher forcing since round 6: the whole chain is bit-identical, so the two
# trajectories cannot drift apart): every counter, the match indices, the LM iterations and the float32 pose of every frame must
# be EQUAL, bit for bit.
COUNTERS = ("frame_id", "n_kp", "n_stereo", "n_match_pass1", "n_match_pass2", "n_lm_edges",
            "n_new_mappoints", "n_local_map", "n_pnp_inliers")

CAM04 = dict(W=1241, H=376, fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104, bf=379.8145)


def _synth():
    return importlib.import_module("stereo_semantic_vo_amd.synth")


def _compare_run(gpu, ref):
    """gpu / ref: lists of (record, cur_mp) of two free-running trackers.  Counters, LM iterations, match indices and the pose
    (CV_32F, 16 floats) identical on every frame."""
    for k, ((res, cur), (rr, rcur)) in enumerate(zip(gpu, ref)):
        for f in COUNTERS + ("lm_iterations",):
            assert res[f] == rr[f], (k, f, int(res[f]), int(rr[f]))
        assert np.array_equal(cur[:rr["n_kp"]], rcur[:rr["n_kp"]]), "frame %d match indices" % k
        assert res["Tcw"].tobytes() == rr["Tcw"].tobytes(), (k, np.abs(res["Tcw"] - rr["Tcw"]).max())


def _track_both(svo, trk, frames, boxes=None, dense=None, per_frame=None):
    """Device frame k, then the oracle's frame k - both free-running.  frames: iterable of (L, R); boxes / dense:
    optional k -> boxes array / dense disparity map for the ORACLE (the device derives its own); per_frame(k): extra checks."""
    gpu, ref = [], []
    for k, (Lk, Rk) in enumerate(frames):
        bx = None if boxes is None else boxes(k)
        res = svo.track_frame(Lk, Rk, boxes=bx) if bx is not None else svo.track_frame(Lk, Rk)
        gpu.append((res.copy(), svo.debug_track_matches()))
        ref.append(trk.track(Lk, Rk, boxes=bx, dense=None if dense is None else dense(k)))
        if per_frame is not None:
            per_frame(k)
    return gpu, ref


# ---------------------------------------------------------------- configs[0], CPU leg (no GPU needed) -------------
def test_config0_kitti04_intrinsics_cpu_plumbing(orc, pkg):
    """The reference's own CPU-runnable case: KITTI04-12 intrinsics, no boxes, through the oracle tracker.  A small
    synthetic sequence rendered with the same camera model must be followed to within the reference's own noise."""
    synth = _synth()
    cam = dict(CAM04, W=640, H=240, cx=320.0, cy=120.0)
    L, R, T = synth.render_sequence(4, cam=cam)
    trk = orc.Tracker(640, 240, dict(fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], bf=cam["bf"]))
    errs = []
    for k in range(4):
        res, _ = trk.track(L[k].numpy(), R[k].numpy())
        Twc = np.linalg.inv(res["Tcw"].reshape(4, 4).astype(np.float64))
        errs.append(np.linalg.norm(Twc[:3, 3] - T[k][:3, 3].numpy()))
        assert res["n_kp"] > 300 and res["n_stereo"] > 150
    trk.close()
    assert errs[0] < 1e-3 and max(errs) < 0.35
    # depth follows bf: the same disparity read with the KITTI00 bf gives a different depth
    assert abs(pkg.KITTI_04_12["bf"] - CAM04["bf"]) < 1e-6 and abs(pkg.KITTI_04_12["fx"] - CAM04["fx"]) < 1e-6
    assert pkg.KITTI_04_12["bf"] != pkg.KITTI_00_02["bf"]


# ---------------------------------------------------------------- configs[0], GPU ---------------------------------
@pytest.fixture(scope="module")
def seq04(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    L, R, T = _synth().render_sequence(12, cam=CAM04, device=torch.device("cuda", 0))
    return L.cpu().numpy(), R.cpu().numpy(), T.numpy()


@pytest.mark.gpu
def test_config0_kitti04_stereo_frame_equals_oracle(pkg, orc, seq04):
    L, R, _ = seq04
    cam = pkg.Camera(**pkg.KITTI_04_12)
    svo = pkg.Svo(1241, 376, max_batch=1)
    g = svo.stereo_frame(L[0], R[0], cam)
    r = orc.stereo_frame(L[0], R[0], cam.bf, cam.fx)
    svo.close()
    assert len(g["kpL"]) == len(r["kpL"]) > 400
    assert g["kpL"].tobytes() == r["kpL"].tobytes()
    assert np.array_equal(g["dL"], r["dL"])
    assert np.array_equal(g["uR"].view(np.uint32), r["uR"].view(np.uint32))
    assert np.array_equal(g["depth"].view(np.uint32), r["depth"].view(np.uint32))
    # the yaml's bf is what scales depth (src/frame.cc:140-164): same disparities, KITTI00 bf -> other depths
    g00 = pkg.Svo(1241, 376, max_batch=1)
    h = g00.stereo_frame(L[0], R[0], pkg.Camera(**pkg.KITTI_00_02))
    g00.close()
    both = (g["depth"] > 0) & (h["depth"] > 0)
    assert both.sum() > 200 and not np.array_equal(g["depth"][both], h["depth"][both])


@pytest.mark.gpu
def test_config0_kitti04_tracked_sequence_equals_oracle(pkg, orc, seq04):
    L, R, T = seq04
    trk = orc.Tracker(1241, 376, pkg.KITTI_04_12)
    svo = pkg.Svo(1241, 376, max_batch=1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_04_12))
    gpu, ref = _track_both(svo, trk, [(L[k], R[k]) for k in range(len(L))])
    trk.close(); svo.close()
    _compare_run(gpu, ref)
    Twc = np.linalg.inv(gpu[-1][0]["Tcw"].reshape(4, 4).astype(np.float64))
    assert np.linalg.norm(Twc[:3, 3] - T[-1][:3, 3]) < 0.6
    assert gpu[-1][0]["n_lm_edges"] > 20


# ---------------------------------------------------------------- configs[4]: boxes + dense ELAS together ----------
def _boxes(k):
    """offline detections in the reference's {left, right, top, bottom} layout (main.cpp:82-95)"""
    return np.array([[500 + 4 * k, 760 + 4 * k, 150, 330], [100, 260, 200, 300 + 2 * k]], np.int32)


@pytest.mark.gpu
def test_config4_boxes_with_dense_elas_depth_equals_oracle(pkg, orc):
    """svo_set_option("depth_source", 1) AND detection boxes in the same frames: the creation gates, F from the
    brute-force matches, the epipolar veto and the dense-map depth lookups all act on one chain."""
    import torch
    from oracle import binding as ob
    if ob.ref_elas_lib() is None:
        pytest.skip("oracle/_ref not built")
    n = 6
    L, R, _ = _synth().render_sequence(n, device=torch.device("cuda", 0))
    L, R = L.cpu().numpy(), R.cpu().numpy()
    trk = orc.Tracker(1241, 376, pkg.KITTI_00_02)
    svo = pkg.Svo(1241, 376, max_batch=1)
    svo.set_option("depth_source", 1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    def gate(k):
        if k > 0:
            F, nv = svo.debug_track_gate()
            assert np.allclose(F.reshape(9), trk.F, rtol=1e-6, atol=1e-9), k
            assert nv == trk.vetoes, (k, nv, trk.vetoes)
    gpu, ref = _track_both(svo, trk, [(L[k], R[k]) for k in range(n)], boxes=_boxes,
                           dense=lambda k: ob.ref_elas(L[k], R[k])[0],   # the reference's own libelas, compiled from its sources
                           per_frame=gate)
    trk.close(); svo.close()
    _compare_run(gpu, ref)
    assert gpu[-1][0]["n_stereo"] > 250
    # the gates bite: fewer map points than the same frames without boxes would create on frame 0
    assert gpu[0][0]["n_new_mappoints"] < gpu[0][0]["n_stereo"]
    # the same configuration through the batched, device-resident entry (svo_elas_batch_dev maps + boxes as HBM arrays)
    dev = torch.device("cuda", 0)
    pitch = 1280
    dL = torch.zeros((n, 376, pitch), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    dL[:, :, :1241] = torch.from_numpy(L).to(dev); dR[:, :, :1241] = torch.from_numpy(R).to(dev)
    bb = np.zeros((n, 2, 4), np.int32)
    for k in range(n):
        bb[k] = _boxes(k)
    tb = torch.from_numpy(bb).to(dev); tn = torch.full((n,), 2, dtype=torch.int32, device=dev)
    res = torch.zeros((n, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    b = pkg.Svo(1241, 376, max_batch=n)
    b.set_option("depth_source", 1)
    b.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    b.track_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, n, res.data_ptr(), boxes=pkg.boxes_dev(tb.data_ptr(), tn.data_ptr(), 2))
    b.sync()
    got = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
    b.close()
    for k in range(n):
        assert got[k].tobytes() == gpu[k][0].tobytes(), k


# ---------------------------------------------------------------- long GPU-vs-oracle runs --------------------------
@pytest.mark.gpu
def test_64_frames_gpu_tracker_equals_oracle(pkg, orc):
    """64 full-size frames, frame by frame against the CPU port (about 13 s of CPU work): culling is active from
    frame 4 on, the pool is compacted 64 times, and pass 2 has to re-evaluate rows whose speculative result went stale
    (the diagnostics in `reserved` prove the path ran)."""
    import torch
    N = 64
    L, R, T = _synth().render_sequence(N, device=torch.device("cuda", 0))
    L, R = L.cpu().numpy(), R.cpu().numpy()
    trk = orc.Tracker(1241, 376, pkg.KITTI_00_02)
    svo = pkg.Svo(1241, 376, max_batch=1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    rounds2 = []
    gpu, ref = _track_both(svo, trk, [(L[k], R[k]) for k in range(N)],
                           per_frame=lambda k: rounds2.append(int(svo.debug_track_frames(0, 1)[0]["rounds"][1])))
    trk.close(); svo.close()
    _compare_run(gpu, ref)
    rec = np.array([g[0] for g in gpu])
    assert rec["n_local_map"][8:].max() > 800                 # four frames of new points
    assert (rec["n_match_pass2"][4:] > 0).sum() > 40          # pass 2 contributes on most frames
    assert (np.array(rounds2) > 1).sum() > 10                 # pass 2 needed more than one round: rows were re-evaluated
    Twc = np.linalg.inv(rec[-1]["Tcw"].reshape(4, 4).astype(np.float64))
    assert np.linalg.norm(Twc[:3, 3] - T[-1][:3, 3].numpy()) < 3.0


@pytest.mark.gpu
def test_unrelated_frames_drive_the_pool_to_its_bound(pkg, orc):
    """Frames that share nothing: no pass ever matches, so every keypoint with depth becomes a new map point on
    every frame.  That is the largest pool the path can reach: the local map holds the points of the last four frames
    (src/Tracking.cc:239-250) plus what the last frame references, <= 5 x 512 rows - the 4096-row pool of the device
    tracker cannot overflow with max_kp <= 512.  GPU and oracle must agree on every counter on the way there."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import util
    W, H = 1241, 376
    n = 8
    trk = orc.Tracker(W, H, pkg.KITTI_00_02)
    svo = pkg.Svo(W, H, max_batch=1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    gpu, ref = [], []
    for k in range(n):
        Lk, Rk = util.shifted_pair(100 + k, W, H, disparity=8 + k)
        ref.append(trk.track(Lk, Rk))
        res = svo.track_frame(Lk, Rk)
        gpu.append((res.copy(), svo.debug_track_matches()))
    trk.close(); svo.close()
    for k, ((res, cur), (rr, rcur)) in enumerate(zip(gpu, ref)):
        for f in COUNTERS:
            assert res[f] == rr[f], (k, f, int(res[f]), int(rr[f]))
        assert np.array_equal(cur[:rr["n_kp"]], rcur[:rr["n_kp"]]), k
    rec = np.array([g[0] for g in gpu])
    assert rec["n_new_mappoints"][1:].min() > 250             # every keypoint with depth creates a point
    assert rec["n_local_map"].max() > 1200
    assert rec["n_local_map"].max() <= 4 * 512


# ---------------------------------------------------------------- a tracked sequence on REAL street images ---------
@pytest.mark.gpu
def test_real_street_images_tracked_sequence_equals_oracle(pkg, orc):
    """The only real stereo data on this machine are the pairs the reference ships with libelas (urban1: 1344 x 391).  A
    KITTI-sized window panned across the pair (12 px per frame, the same window in both images: real texture, real stereo
    disparities, a steady image motion) gives the tracker eight consecutive real frames: descriptors, Hamming thresholds
    15 / 30, the ratio test and the map-point lifecycle then act on real street texture instead of the synthetic scene.
    Device tracker (bit-comparable RANSAC mode) against the oracle: everything exact, pose to BASELINE.md's tolerance."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import util
    frames = [util.urban_pair(1241, 376, x0=4 + 12 * k, y0=8) for k in range(8)]
    trk = orc.Tracker(1241, 376, pkg.KITTI_00_02)
    svo = pkg.Svo(1241, 376, max_batch=1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    gpu, ref = _track_both(svo, trk, frames)
    trk.close(); svo.close()
    _compare_run(gpu, ref)
    rec = np.array([g[0] for g in gpu])
    assert rec["n_kp"].min() > 400 and rec["n_stereo"].min() > 150          # real texture: plenty of corners and stereo matches
    assert (rec["n_match_pass1"][1:] > 10).all() and rec["n_lm_edges"][1:].min() >= 10
