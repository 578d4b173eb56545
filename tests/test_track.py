"""Tracking loop (Tracking::Track, reference src/Tracking.cc:180-252).
CPU: the oracle tracker on a short synthetic sequence against exact ground truth.
GPU (-m gpu): the device-resident tracker against the oracle tracker on the same frames -
match indices bit-exact, poses within the stated tolerance."""
import importlib

import numpy as np
import pytest

N_FRAMES = 7
POSE_TOL_T = 5e-4     # metres per frame   (far inside the reference's noise; see tests/test_configs.py)
POSE_TOL_R = 5e-5     # rotation-matrix entries (~radians)


@pytest.fixture(scope="module")
def sequence(pkg):
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    L, R, T = synth.render_sequence(N_FRAMES)
    return L.numpy(), R.numpy(), T.numpy()


@pytest.fixture(scope="module")
def oracle_run(orc, pkg, sequence):
    L, R, T = sequence
    trk = orc.Tracker(L.shape[2], L.shape[1], pkg.KITTI_00_02)
    out = [trk.track(L[k], R[k]) for k in range(N_FRAMES)]
    trk.close()
    return out


def test_synth_geometry(sequence):
    L, R, T = sequence
    assert L.shape == (N_FRAMES, 376, 1241) and L.dtype == np.uint8
    step = np.linalg.norm(np.diff(T[:, :3, 3], axis=0), axis=1)
    assert np.allclose(step, 1.0, atol=1e-9)
    assert 20 < L[0].std() < 40 and not np.array_equal(L[0], R[0])


def test_oracle_tracker_follows_ground_truth(oracle_run, sequence):
    _, _, T = sequence
    errs = []
    for k, (res, cur) in enumerate(oracle_run):
        Twc = np.linalg.inv(res["Tcw"].reshape(4, 4).astype(np.float64))
        errs.append(np.linalg.norm(Twc[:3, 3] - T[k][:3, 3]))
        assert res["frame_id"] == k and res["n_kp"] > 400 and res["n_stereo"] > 250
        if k > 0:
            assert res["n_lm_edges"] == res["n_match_pass1"] + res["n_match_pass2"] >= 20
            assert res["n_pnp_inliers"] >= 15
    assert errs[0] < 1e-3
    assert max(errs) < 0.35          # the reference itself drifts ~0.1 m/frame
    steps = [np.linalg.norm(np.linalg.inv(oracle_run[k][0]["Tcw"].reshape(4, 4).astype(np.float64))[:3, 3] -
                            np.linalg.inv(oracle_run[k - 1][0]["Tcw"].reshape(4, 4).astype(np.float64))[:3, 3])
             for k in range(1, N_FRAMES)]
    assert all(0.8 < s < 1.2 for s in steps)


def test_local_map_window(oracle_run):
    # points older than 3 frames leave the local map (src/Tracking.cc:239-250)
    sizes = [r["n_local_map"] for r, _ in oracle_run]
    assert sizes[3] > sizes[0] and sizes[5] <= sizes[3] + 60


def test_oracle_tail_entry_equals_full_frame_entry(orc, pkg, sequence, oracle_run):
    """orc_track_tail fed with the oracle's own front-end results == orc_track_frame on the images, record for record:
    the split the full-length GPU parity test relies on (it feeds the tail with the device front end's outputs)."""
    L, R, _ = sequence
    cam = pkg.KITTI_00_02
    trk = orc.Tracker(L.shape[2], L.shape[1], cam)
    for k in range(N_FRAMES):
        fe = orc.stereo_frame(L[k], R[k], cam["bf"], cam["fx"])
        res, cur, pnp, Tp = trk.track_tail(fe["kpL"], fe["dL"], fe["depth"])
        ref, ref_cur = oracle_run[k]
        assert res.tobytes() == ref.tobytes(), k
        assert np.array_equal(cur, ref_cur), k
        if k > 0:
            assert pnp["ok"] == 1 and pnp["n_inliers"] == ref["n_pnp_inliers"] and 0 <= pnp["best_hypothesis"] < pnp["iterations"] <= 100
            assert np.abs(Tp[:3, 3] - ref["Tcw"].reshape(4, 4)[:3, 3]).max() < 0.5       # a five-point minimal solution: the LM moves it by centimetres
    trk.close()


def test_oracle_tail_teacher_forcing(orc, pkg, sequence, oracle_run):
    """Tcw_force: the frame's own pose is still reported, but the chain continues from the forced pose - forcing the
    oracle's own poses changes nothing; forcing a shifted pose moves the NEXT frame's result, not this one's."""
    L, R, _ = sequence
    cam = pkg.KITTI_00_02
    fes = [orc.stereo_frame(L[k], R[k], cam["bf"], cam["fx"]) for k in range(4)]
    a = orc.Tracker(L.shape[2], L.shape[1], cam)
    b = orc.Tracker(L.shape[2], L.shape[1], cam)
    for k in range(4):
        ref = oracle_run[k][0]
        ra = a.track_tail(fes[k]["kpL"], fes[k]["dL"], fes[k]["depth"], Tcw_force=ref["Tcw"])[0]
        assert ra.tobytes() == ref.tobytes(), k
        shifted = ref["Tcw"].copy()
        shifted[3] += 0.5 if k == 1 else 0.0                     # tx of frame 1 moved by half a metre
        rb = b.track_tail(fes[k]["kpL"], fes[k]["dL"], fes[k]["depth"], Tcw_force=shifted)[0]
        if k <= 1:
            assert rb.tobytes() == ref.tobytes(), k
        else:
            # points created at the end of frame 1 sit 0.5 m off: the counters (pose-free) stay, the pose moves
            for f in ("n_kp", "n_stereo", "n_match_pass1", "n_match_pass2", "n_lm_edges", "n_new_mappoints", "n_local_map"):
                assert rb[f] == ref[f], (k, f)
    assert np.abs(rb["Tcw"] - ref["Tcw"]).max() > 1e-3
    a.close(); b.close()


def _check_frame(k, res, cur, ref, ref_cur, exact):
    """One tracked frame against the (free-running) oracle.  exact (svo_set_option "epnp_exact" 2 = the default, order-preserving
    wave EPnP; 1 = its sequential checker): everything pinned - RANSAC consensus, LM iterations, the CV_32F pose BIT FOR BIT.
    Mode 0 (the statistical wave solver, an option): the index chain is still exact; the pose chain's RANSAC samples are
    solved with another rounding (see tests/test_full_length.py), hence the bands."""
    for f in ("frame_id", "n_kp", "n_stereo", "n_match_pass1", "n_match_pass2", "n_lm_edges", "n_new_mappoints", "n_local_map"):
        assert res[f] == ref[f], (k, f, res[f], ref[f])
    assert np.array_equal(cur[:ref["n_kp"]], ref_cur[:ref["n_kp"]]), "frame %d match indices" % k
    T, Tr = res["Tcw"].reshape(4, 4), ref["Tcw"].reshape(4, 4)
    if exact:
        assert res["n_pnp_inliers"] == ref["n_pnp_inliers"] and res["lm_iterations"] == ref["lm_iterations"], k
        assert res["Tcw"].tobytes() == ref["Tcw"].tobytes(), (k, np.abs(T - Tr).max())
    else:
        assert abs(int(res["n_pnp_inliers"]) - int(ref["n_pnp_inliers"])) <= max(2, 0.1 * int(ref["n_lm_edges"])), k
        assert abs(int(res["lm_iterations"]) - int(ref["lm_iterations"])) <= 1, k
        assert np.abs(T[:3, 3] - Tr[:3, 3]).max() < POSE_TOL_T, k
        assert np.abs(T[:3, :3] - Tr[:3, :3]).max() < POSE_TOL_R, k


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [2, 1, 0])
@pytest.mark.parametrize("lcap,nblk", [(8, 3), (1, 3), (8, 0), (2, 1)])
def test_gpu_tracker_matches_oracle(pkg, sequence, oracle_run, lcap, nblk, exact):
    """(8, 3) is the product's configuration; the others force the matching passes onto their fall-back paths - rows
    with more claimable keypoints than packed entries (evaluated on the full distance row) and entries whose stored
    runner-up blockers do not suffice (looked up in the row)."""
    L, R, _ = sequence
    svo = pkg.Svo(L.shape[2], L.shape[1], max_batch=1)
    svo.set_option("track_lcap", lcap)
    svo.set_option("track_nblk", nblk)
    svo.set_option("epnp_exact", exact)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    for k in range(N_FRAMES):
        res = svo.track_frame(L[k], R[k])
        cur = svo.debug_track_matches()
        ref, ref_cur = oracle_run[k]
        _check_frame(k, res, cur, ref, ref_cur, exact)
    svo.close()


@pytest.mark.gpu
def test_gpu_batch_tracker_equals_frame_by_frame(pkg, sequence):
    """svo_track_batch_dev over HBM-resident frames == svo_track_frame one by one."""
    import torch
    L, R, _ = sequence
    H, W = L.shape[1], L.shape[2]
    cam = pkg.Camera(**pkg.KITTI_00_02)
    a = pkg.Svo(W, H, max_batch=1)
    a.track_reset(cam)
    single = [a.track_frame(L[k], R[k]) for k in range(N_FRAMES)]
    a.close()
    pitch = 1280
    dev = torch.device("cuda", 0)
    dL = torch.zeros((N_FRAMES, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    dL[:, :, :W] = torch.from_numpy(L).to(dev); dR[:, :, :W] = torch.from_numpy(R).to(dev)
    res = torch.zeros((N_FRAMES, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # the pose chain as ONE launch per frame (k_tp_tail_ord, the default) and as two (RANSAC samples, then the frame part)
    for fused in (1, 0):
        b = pkg.Svo(W, H, max_batch=N_FRAMES)
        b.set_option("tail_fused", fused)
        b.track_reset(cam)
        b.track_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, N_FRAMES, res.data_ptr())
        b.sync()
        out = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
        assert b.track_overflowed() == 0
        b.close()
        for k in range(N_FRAMES):
            assert out[k].tobytes() == single[k].tobytes(), (fused, k)


# ---- depth source 1: dense ELAS map (BASELINE configs[4] without YOLO; reference data flow src/Tracking.cc:226-228) ----
def _ref_elas_or_skip():
    from oracle import binding as ob
    if ob.ref_elas_lib() is None:
        pytest.skip("oracle/_ref not built")
    return ob


@pytest.fixture(scope="module")
def oracle_run_dense(orc, pkg, sequence):
    """Oracle tracker fed with the disparity maps of the REAL reference libelas (oracle/_ref)."""
    ob = _ref_elas_or_skip()
    L, R, T = sequence
    trk = orc.Tracker(L.shape[2], L.shape[1], pkg.KITTI_00_02)
    out = [trk.track(L[k], R[k], dense=ob.ref_elas(L[k], R[k])[0]) for k in range(N_FRAMES)]
    trk.close()
    return out


def test_oracle_tracker_with_dense_reference_depth_follows_ground_truth(oracle_run_dense, sequence):
    _, _, T = sequence
    errs = []
    for k, (res, cur) in enumerate(oracle_run_dense):
        Twc = np.linalg.inv(res["Tcw"].reshape(4, 4).astype(np.float64))
        errs.append(np.linalg.norm(Twc[:3, 3] - T[k][:3, 3]))
        assert res["n_kp"] > 400 and res["n_stereo"] > 250
    assert errs[0] < 1e-3 and max(errs) < 0.5   # integer-ish ELAS disparities: coarser depth than the sub-pixel sparse matcher


@pytest.mark.gpu
def test_gpu_tracker_with_dense_elas_depth_matches_oracle(pkg, sequence, oracle_run_dense):
    """svo_set_option("depth_source", 1): ORB (left) + svo_elas on the device + depth lookups, against the
    oracle tracker that reads the reference libelas' own maps."""
    L, R, _ = sequence
    svo = pkg.Svo(L.shape[2], L.shape[1], max_batch=1)
    svo.set_option("depth_source", 1)
    svo.set_option("epnp_exact", 1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    for k in range(N_FRAMES):
        res = svo.track_frame(L[k], R[k])
        cur = svo.debug_track_matches()
        ref, ref_cur = oracle_run_dense[k]
        _check_frame(k, res, cur, ref, ref_cur, True)
    svo.close()


# ---- depth source 2: dense MSA map, the reference's live configuration (src/Tracking.cc:225-228, frame::MB) ----
MSA_FRAMES = 3


def _as_bgr(g):
    return np.ascontiguousarray(np.repeat(np.asarray(g)[:, :, None], 3, 2))


@pytest.mark.gpu
def test_gpu_tracker_with_dense_msa_depth_matches_oracle(orc, pkg, sequence):
    """svo_set_option("depth_source", 2): ORB (left) + MSA::solve(l, r, 48, 1) on the device + depth lookups, against
    the oracle tracker reading the maps of the CPU restatement of MSA (frame by frame, single and batched call)."""
    import torch
    from oracle import binding as ob
    L, R, _ = sequence
    H, W = L.shape[1], L.shape[2]
    trk = orc.Tracker(W, H, pkg.KITTI_00_02)
    svo = pkg.Svo(W, H, max_batch=MSA_FRAMES)
    svo.set_option("depth_source", 2)
    svo.set_option("epnp_exact", 1)
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    single = []
    for k in range(MSA_FRAMES):
        dmap = ob.msa_solve(_as_bgr(L[k]), _as_bgr(R[k]), 48, 1).astype(np.float32)
        ref, ref_cur = trk.track(L[k], R[k], dense=dmap)
        res = svo.track_frame(L[k], R[k])
        cur = svo.debug_track_matches()
        _check_frame(k, res, cur, ref, ref_cur, True)
        assert res["n_stereo"] > 100
        single.append(res.copy())
    trk.close()
    # the batched entry point walks the same frames
    pitch = 1280
    dev = torch.device("cuda", 0)
    dL = torch.zeros((MSA_FRAMES, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    dL[:, :, :W] = torch.as_tensor(np.asarray(L[:MSA_FRAMES])).to(dev); dR[:, :, :W] = torch.as_tensor(np.asarray(R[:MSA_FRAMES])).to(dev)
    out = torch.zeros(MSA_FRAMES * pkg.TRACK_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    svo.track_reset(pkg.Camera(**pkg.KITTI_00_02))
    svo.track_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, MSA_FRAMES, out.data_ptr())
    svo.sync()
    rec = out.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
    for k in range(MSA_FRAMES):
        assert rec[k].tobytes() == single[k].tobytes(), k
    svo.close()


@pytest.mark.gpu
def test_many_sequences_staged_ransac_launches_equal_single_chains(pkg, sequence):
    """With eight or more sequences per step the RANSAC samples go out in three launches (F, F, the rest; option "hyp_first")
    whose later workgroups leave when cv::solvePnPRansac's iteration bound says the loop never reaches their sample: whatever F
    is, every sequence's records must equal a single chain's (which computes all 100 samples)."""
    import torch
    L, R, _ = sequence
    H, W = L.shape[1], L.shape[2]
    S, STEPS = 8, 4                       # sequence q starts at frame q % 3
    assert N_FRAMES >= 2 + STEPS
    cam = pkg.Camera(**pkg.KITTI_00_02)
    pitch = 1280
    dev = torch.device("cuda", 0)
    dL = torch.zeros((N_FRAMES, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    dL[:, :, :W] = torch.from_numpy(L).to(dev); dR[:, :, :W] = torch.from_numpy(R).to(dev)
    rec = pkg.TRACK_DTYPE.itemsize
    fb = H * pitch
    singles = {}
    for q in (0, 1, 2):
        a = pkg.Svo(W, H, max_batch=STEPS)
        a.track_reset(cam)
        res = torch.zeros((STEPS, rec), dtype=torch.uint8, device=dev)
        a.track_batch_dev(dL.data_ptr() + q * fb, dR.data_ptr() + q * fb, pitch, STEPS, res.data_ptr())
        a.sync()
        singles[q] = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
        a.close()
    # step t's pairs, one per sequence, side by side
    idx = torch.tensor([[(q % 3) + t for q in range(S)] for t in range(STEPS)], device=dev)
    mL = dL[idx.reshape(-1)].reshape(STEPS, S, H, pitch).contiguous()
    mR = dR[idx.reshape(-1)].reshape(STEPS, S, H, pitch).contiguous()
    torch.cuda.synchronize()
    m = pkg.Svo(W, H, max_batch=S)
    with pytest.raises(pkg.SvoError):
        m.set_option("hyp_first", 6)          # multiples of four only (the statistical solver's workgroups hold four samples)
    for first in (8, 4, 16, -2):
        # (-2: option "tail_semi" = 2 instead - a step's pose chains as two launches, the first 8 samples of every sequence, then the
        # other 92 (which replay the bound and leave) with the frame parts)
        if first > 0:
            m.set_option("hyp_first", first)
        else:
            m.set_option("hyp_first", 8)
            m.set_option("tail_semi", 2)
        m.track_multi_reset(S, cam)
        out = torch.zeros((STEPS, S, rec), dtype=torch.uint8, device=dev)
        for t in range(STEPS):
            m.track_multi_step_dev(mL[t].data_ptr(), mR[t].data_ptr(), pitch, S, out[t].data_ptr())
        m.sync()
        assert m.track_overflowed() == 0
        multi = out.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(STEPS, S)
        for q in range(S):
            for t in range(STEPS):
                assert multi[t, q].tobytes() == singles[q % 3][t].tobytes(), (first, q, t)
    m.close()


@pytest.mark.gpu
def test_multi_sequence_tracker_equals_independent_chains(pkg, sequence):
    """svo_track_multi_step_dev: S staggered sequences advanced together == S single-sequence trackers,
    record for record (byte-identical svo_track_result)."""
    import torch
    L, R, _ = sequence
    H, W = L.shape[1], L.shape[2]
    S, STEPS = 3, N_FRAMES - 2            # sequence q sees frames q, q+1, ...
    cam = pkg.Camera(**pkg.KITTI_00_02)
    pitch = 1280
    dev = torch.device("cuda", 0)
    dL = torch.zeros((N_FRAMES, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    dL[:, :, :W] = torch.from_numpy(L).to(dev); dR[:, :, :W] = torch.from_numpy(R).to(dev)
    rec = pkg.TRACK_DTYPE.itemsize
    fb = H * pitch
    singles = []
    for q in range(S):
        a = pkg.Svo(W, H, max_batch=STEPS)
        a.track_reset(cam)
        res = torch.zeros((STEPS, rec), dtype=torch.uint8, device=dev)
        a.track_batch_dev(dL.data_ptr() + q * fb, dR.data_ptr() + q * fb, pitch, STEPS, res.data_ptr())
        a.sync()
        singles.append(res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1))
        a.close()
    m = pkg.Svo(W, H, max_batch=S)
    m.track_multi_reset(S, cam)
    out = torch.zeros((STEPS, S, rec), dtype=torch.uint8, device=dev)
    for t in range(STEPS):                # pairs t .. t+S-1 are the next frames of sequences 0 .. S-1
        m.track_multi_step_dev(dL.data_ptr() + t * fb, dR.data_ptr() + t * fb, pitch, S, out[t].data_ptr())
    m.sync()
    multi = out.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(STEPS, S)
    for q in range(S):
        for t in range(STEPS):
            assert multi[t, q].tobytes() == singles[q][t].tobytes(), (q, t)
    assert multi[-1, 0]["n_lm_edges"] > 20
    # pipelined steps (the front end of step t + 1 beside the tail of step t, alternating output sets): same records,
    # also across a reset in the middle of a run
    m.set_option("multi_pipeline", 1)
    for rep in range(2):
        m.track_multi_reset(S, cam)
        out2 = torch.zeros((STEPS, S, rec), dtype=torch.uint8, device=dev)
        for t in range(STEPS):
            m.track_multi_step_dev(dL.data_ptr() + t * fb, dR.data_ptr() + t * fb, pitch, S, out2[t].data_ptr())
        m.sync()
        assert out2.cpu().numpy().tobytes() == out.cpu().numpy().tobytes(), rep
    m.set_option("multi_pipeline", 0)
    with pytest.raises(pkg.SvoError):     # single-sequence entry points refuse a multi-sequence state
        m.track_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, 1, out.data_ptr())
    with pytest.raises(pkg.SvoError):     # a step must advance exactly the sequences that were reset
        m.track_multi_step_dev(dL.data_ptr(), dR.data_ptr(), pitch, S - 1, out.data_ptr())
    with pytest.raises(pkg.SvoError):     # more sequences than the context was created for
        m.track_multi_reset(S + 1, cam)
    m.close()


@pytest.mark.gpu
def test_gpu_batch_tracker_with_dense_depth_equals_frame_by_frame(pkg, sequence):
    """depth_source = 1 in the batched, device-resident mode (svo_elas_batch_dev under svo_track_batch_dev) ==
    svo_track_frame one by one with the same option."""
    import torch
    L, R, _ = sequence
    H, W = L.shape[1], L.shape[2]
    cam = pkg.Camera(**pkg.KITTI_00_02)
    a = pkg.Svo(W, H, max_batch=1)
    a.set_option("depth_source", 1)
    a.track_reset(cam)
    single = [a.track_frame(L[k], R[k]) for k in range(N_FRAMES)]
    a.close()
    pitch = 1280
    dev = torch.device("cuda", 0)
    dL = torch.zeros((N_FRAMES, H, pitch), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    dL[:, :, :W] = torch.from_numpy(L).to(dev); dR[:, :, :W] = torch.from_numpy(R).to(dev)
    res = torch.zeros((N_FRAMES, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    b = pkg.Svo(W, H, max_batch=N_FRAMES)
    b.set_option("depth_source", 1)
    b.track_reset(cam)
    b.track_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, N_FRAMES, res.data_ptr())
    b.sync()
    out = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
    b.close()
    assert out[-1]["n_stereo"] > 250 and out[-1]["n_lm_edges"] > 20
    for k in range(N_FRAMES):
        assert out[k].tobytes() == single[k].tobytes(), k


@pytest.mark.gpu
def test_tracking_is_independent_of_how_the_sequence_is_batched(pkg):
    """64 frames of 1241x376 through svo_track_batch_dev in one call, in 4 calls of 16 and in 64 calls of 1:
    byte-identical records (the chain is the same whatever the batching), and two identical runs agree."""
    import importlib
    import torch
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    N = 64
    dev = torch.device("cuda", 0)
    L, R, T = synth.render_sequence(N, device=dev)      # rendered on the GPU (the CPU renderer takes ~3 s per frame)
    H, W = L.shape[1], L.shape[2]
    pitch = 1280
    dL = torch.zeros((N, H, pitch), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    dL[:, :, :W] = L.to(dev); dR[:, :, :W] = R.to(dev)
    cam = pkg.Camera(**pkg.KITTI_00_02)
    rec = pkg.TRACK_DTYPE.itemsize
    fb = H * pitch

    def run(chunk, **opts):
        s = pkg.Svo(W, H, max_batch=chunk)
        for key, v in opts.items():
            s.set_option(key, v)
        s.track_reset(cam)
        res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for off in range(0, N, chunk):
            s.track_batch_dev(dL.data_ptr() + off * fb, dR.data_ptr() + off * fb, pitch, chunk, res.data_ptr() + off * rec)
        s.sync()
        assert s.track_overflowed() == 0, ("capacity / lost hand-over flag", opts)
        out = res.cpu().numpy().tobytes()
        s.close()
        return out

    a = run(64)
    assert a == run(64) and a == run(16) and a == run(1)
    # strategy switches never change a record: one packed entry per row (everything "dense"), no stored blockers, the
    # pose chain taking frames over one by one or sixteen at a time
    assert a == run(64, track_lcap=1) and a == run(64, track_nblk=0) and a == run(64, track_group=1) and a == run(64, track_group=16)
    # the hand-over between the two chains: stream events per group (default) or per-frame tags polled by the pose kernels
    assert a == run(64, pose_flag=1) and a == run(16, pose_flag=1) and a == run(1, pose_flag=1, track_group=1)
    # another share of the CUs for the front end's stream: that stream is destroyed and made again through the stream picker
    # (the other three of the context's four hardware queues stay where they are) - same records
    assert a == run(16, fe_cu_percent=25) and a == run(16, fe_cu_percent=100)
    r = np.frombuffer(a, pkg.TRACK_DTYPE)
    Twc = np.linalg.inv(r[-1]["Tcw"].reshape(4, 4).astype(np.float64))
    assert np.linalg.norm(Twc[:3, 3] - T[-1][:3, 3].cpu().numpy()) < 3.0      # 63 m path, no loop closing


@pytest.mark.gpu
def test_a_sample_that_never_reports_makes_its_frame_a_pnp_failure_not_a_stale_pose(pkg, sequence):
    """The fused pose launch's frame workgroup waits (bounded) for sample workgroups of its own launch.  With the test switch
    "debug_lose_sample" one of the first eight samples never reports: the frame must be tracked as cv::solvePnPRansac
    returning false (the last pose stays, src/pnpmatch.cc:227 with an unusable result), its record marked n_pnp_inliers = -1,
    svo_sync must return SVO_E_TIMEOUT once, and the context must carry on with the two-launch pose chain - frames before
    the event identical to an undisturbed run, frames after it tracked normally (no silently wrong pose)."""
    L, R, T = sequence
    H, W = L.shape[1], L.shape[2]
    cam = pkg.Camera(**pkg.KITTI_00_02)
    ref = pkg.Svo(W, H)
    ref.track_reset(cam)
    want = [ref.track_frame(L[k], R[k]) for k in range(N_FRAMES)]
    ref.close()
    ctx = pkg.Svo(W, H)
    ctx.track_reset(cam)
    got = []
    for k in range(N_FRAMES):
        if k == 3:
            ctx.set_option("debug_lose_sample", 3)          # sample 2 of frame 3's launch
            with pytest.raises(pkg.SvoError, match="timed out"):
                ctx.track_frame(L[k], R[k])
            ctx.set_option("debug_lose_sample", 0)
            assert ctx.track_overflowed() == 4
            got.append(None)
            continue
        got.append(ctx.track_frame(L[k], R[k]))
    for k in range(3):
        assert got[k].tobytes() == want[k].tobytes(), k
    # the frames after the event: tracked (PnP consensus found, a pose within a few centimetres of the undisturbed run's)
    for k in range(4, N_FRAMES):
        assert got[k]["n_pnp_inliers"] > 4 and got[k]["frame_id"] == k
        c_got = -got[k]["Tcw"].reshape(4, 4)[:3, :3].T @ got[k]["Tcw"].reshape(4, 4)[:3, 3]
        c_want = -want[k]["Tcw"].reshape(4, 4)[:3, :3].T @ want[k]["Tcw"].reshape(4, 4)[:3, 3]
        assert np.linalg.norm(c_got - c_want) < 0.1, k
    ctx.sync()                                              # (reported once; the context stays usable)
    ctx.close()


@pytest.mark.gpu
def test_dense_stage_pose_chain_in_two_launches_equals_the_separate_launches(pkg):
    """depth_source = 1, one sequence ("tail_semi", default 1): the frame's first 8 RANSAC samples go out as one launch, the other 92
    and the frame part as a second one whose sample workgroups replay cv::solvePnPRansac's iteration bound and leave (or, beyond
    sample 16, first wait for samples 8..15 of their own launch).  Records and RANSAC outcomes must equal the chain with all 100
    samples solved in a launch of their own ("tail_semi" = 0) - on a sequence cut so that some frames need more than 8 and more
    than 16 samples (every third frame of the synthetic drive: three times the motion between frames)."""
    import importlib
    import torch
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    dev = torch.device("cuda", 0)
    n_all = 72
    L, R, _ = synth.render_sequence(n_all, device=dev)
    idx = torch.tensor([0, 1, 2, 3] + list(range(6, n_all, 3)), device=dev)
    n = int(idx.numel())
    H, W = int(L.shape[1]), int(L.shape[2])
    pitch = 1280
    dL = torch.zeros((n, H, pitch), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    dL[:, :, :W] = L[idx]; dR[:, :, :W] = R[idx]
    cam = pkg.Camera(**pkg.KITTI_00_02)
    rec = pkg.TRACK_DTYPE.itemsize
    runs = {}
    for semi in (0, 1):
        ctx = pkg.Svo(W, H, max_batch=n)
        ctx.set_option("depth_source", 1)
        ctx.set_option("tail_semi", semi)
        ctx.track_reset(cam)
        res = torch.zeros((n, rec), dtype=torch.uint8, device=dev)
        ctx.track_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, n, res.data_ptr())
        ctx.sync()
        assert ctx.track_overflowed() == 0
        dbg = ctx.debug_track_frames(0, n)
        runs[semi] = (res.cpu().numpy().tobytes(), dbg["pnp_best"].copy(), dbg["pnp_iterations"].copy(), dbg["pnp_inliers"].copy())
        ctx.close()
    it = runs[0][2]
    assert (it > 8).any() and (it > 16).any(), it          # the sequence exercises all three kinds of sample workgroup
    assert (it <= 8).sum() >= 4
    for k in range(1, 4):
        assert np.array_equal(runs[0][k], runs[1][k]), k
    assert runs[0][0] == runs[1][0]
