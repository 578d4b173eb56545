"""BASELINE configs[2] at its FULL length: every one of the 4,541 frames of the synth-kitti00 run that bench.py times
meets the CPU restatement - not a 64-frame prefix.

The device front end (bit-exact against the oracle's by tests/test_gpu_parity.py) produces keypoints, descriptors and
depths for all frames; the device tail (svo_track_tail_dev) and the oracle tail (orc_track_tail, reference
src/Tracking.cc:231-250 and what it calls) then walk the same 4,541 front-end results:

  * index chain, EXACT on every frame: all counters, and the IDENTITY (creation sequence number) of the map point behind
    every match and every created point - 4,541 x 500 integers.  The run takes the map-point ids past 2^20 (the device's
    position table is a ring over ids: it wraps around frame ~3,300) and contains frames whose greedy passes need more
    than 30 resolution rounds;
  * pose chain, per frame under TEACHER FORCING: the oracle computes its own PnP + LM pose for frame k from the same
    map-point positions the device used (it continues from the device's pose after every frame), so each frame is an
    independent comparison on identical inputs and the tolerance does not have to grow along the path:
    BASELINE.md's 1e-4 m / 1e-5 rad in the DEFAULT mode ("epnp_exact" = 2, OpenCV's operations with their rounding over a
    wavefront per RANSAC sample) and in its sequential checker (mode 1) (measured: 1e-6 m, one float32 ulp), RANSAC
    winner / visited samples / consensus / LM iterations identical there; the statistical wave solver (mode 0, an
    option) is validated statistically against the same oracle: same discrete outcome on >= 95 % of the frames, the
    BASELINE pose tolerance on >= 94 %;
  * a second, free-running oracle (no forcing) gives the ATE between the two whole trajectories.
"""
import importlib
import json
import os
import threading

import numpy as np
import pytest

W, H, PITCH, K = 1241, 376, 1280, 500
N_FULL = int(os.environ.get("SVO_FULL_FRAMES", "4541"))
COUNTERS = ("frame_id", "n_kp", "n_stereo", "n_match_pass1", "n_match_pass2", "n_lm_edges", "n_new_mappoints", "n_local_map")
TOL_T, TOL_R = 1e-4, 1e-5            # metres / rotation-matrix entries (~rad), per frame


@pytest.fixture(scope="module")
def run(pkg):
    """Front end of all frames on the device, results on the host; the resident device arrays for the tails."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    dev = torch.device("cuda", 0)
    N = N_FULL
    cam = pkg.Camera(**pkg.KITTI_00_02)
    kp = torch.zeros((N, K, pkg.KP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    desc = torch.zeros((N, K, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(N, dtype=torch.int32, device=dev)
    depth = torch.zeros((N, K), dtype=torch.float32, device=dev)
    fe = pkg.Svo(W, H, max_batch=64)
    Ts = []
    for c0 in range(0, N, 64):
        c = min(64, N - c0)
        L, R, T = synth.render_sequence(c, seed=synth.BASE_SEED, device=dev, start=c0)
        dL = torch.zeros((c, H, PITCH), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
        dL[:, :, :W] = L; dR[:, :, :W] = R
        torch.cuda.synchronize()
        fe.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), PITCH, c, cam, d_kpL=kp[c0:].data_ptr(), d_descL=desc[c0:].data_ptr(),
                              d_nL=n[c0:].data_ptr(), d_depth=depth[c0:].data_ptr())
        fe.sync()
        Ts.append(T.cpu())
    fe.close()
    host = dict(kp=kp.cpu().numpy().view(pkg.KP_DTYPE).reshape(N, K), desc=desc.cpu().numpy(), n=n.cpu().numpy(),
                depth=depth.cpu().numpy(), T_gt=torch.cat(Ts).numpy())
    return dict(N=N, cam=cam, kp=kp, desc=desc, n=n, depth=depth, host=host)


def _device_tail(pkg, run, epnp_mode):
    import torch
    N = run["N"]
    dev = run["kp"].device
    rec = pkg.TRACK_DTYPE.itemsize
    tail = pkg.Svo(W, H, max_batch=1)
    tail.set_option("epnp_exact", epnp_mode)
    tail.track_reset(run["cam"])
    res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
    dbg = []
    CH = 512
    for c0 in range(0, N, CH):
        c = min(CH, N - c0)
        tail.track_tail_dev(run["kp"][c0:].data_ptr(), run["desc"][c0:].data_ptr(), run["n"][c0:].data_ptr(),
                            run["depth"][c0:].data_ptr(), K, c, res.data_ptr() + c0 * rec)
        tail.sync()
        dbg.append(tail.debug_track_frames(0, c))
    flag = tail.track_overflowed()
    print("EPNP_FALLBACKS mode %d: %d samples" % (epnp_mode, tail.track_epnp_fallbacks()))
    tail.close()
    return res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1), np.concatenate(dbg), flag


def _rel(a, b):
    return b.reshape(4, 4).astype(np.float64) @ np.linalg.inv(a.reshape(4, 4).astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ord", "exact", "fast"])
def test_every_frame_of_the_kitti00_sized_run_against_the_oracle(pkg, orc, run, mode):
    N, h = run["N"], run["host"]
    gpu, dbg, flag = _device_tail(pkg, run, {"ord": 2, "exact": 1, "fast": 0}[mode])
    assert flag == 0, "device tracker reported a capacity overflow"
    camd = pkg.KITTI_00_02
    forced = orc.Tracker(W, H, camd)
    free = orc.Tracker(W, H, camd)
    free_poses = []

    def free_run():                      # the free-running oracle on its own thread (ctypes releases the GIL)
        for k in range(N):
            nk = int(h["n"][k])
            free_poses.append(free.track_tail(h["kp"][k, :nk], h["desc"][k, :nk], h["depth"][k, :nk])[0]["Tcw"].copy())
    th = threading.Thread(target=free_run)
    th.start()
    dt, dr, d_inl, d_it, winners_differ, iters_differ = [], [], [], [], 0, 0
    dpnp = []
    for k in range(N):
        nk = int(h["n"][k])
        rr, cur, pnp, Tp = forced.track_tail(h["kp"][k, :nk], h["desc"][k, :nk], h["depth"][k, :nk], Tcw_force=gpu[k]["Tcw"])
        g = gpu[k]
        for f in COUNTERS:
            assert g[f] == rr[f], (k, f, int(g[f]), int(rr[f]))
        assert np.array_equal(dbg[k]["match_gid"][:nk], forced.match_gid[:nk]), "frame %d: matched map-point identities" % k
        assert np.array_equal(dbg[k]["new_gid"][:nk], forced.new_gid[:nk]), "frame %d: created map-point identities" % k
        T, Tr = g["Tcw"].reshape(4, 4), rr["Tcw"].reshape(4, 4)
        dt.append(float(np.abs(T[:3, 3] - Tr[:3, 3]).max())); dr.append(float(np.abs(T[:3, :3] - Tr[:3, :3]).max()))
        d_inl.append(abs(int(g["n_pnp_inliers"]) - int(rr["n_pnp_inliers"])))
        d_it.append(abs(int(g["lm_iterations"]) - int(rr["lm_iterations"])))
        if k > 0 and pnp["ok"]:
            winners_differ += int(dbg[k]["pnp_best"]) != pnp["best_hypothesis"]
            iters_differ += int(dbg[k]["pnp_iterations"]) != pnp["iterations"]
            dpnp.append(float(np.abs(dbg[k]["T_pnp"].reshape(4, 4)[:3, 3] - Tp[:3, 3]).max()))
    th.join()
    forced.close(); free.close()
    dt, dr, d_inl, d_it, dpnp = map(np.array, (dt, dr, d_inl, d_it, dpnp))
    ca = np.array([np.linalg.inv(gpu[k]["Tcw"].reshape(4, 4).astype(np.float64))[:3, 3] for k in range(N)])
    cb = np.array([np.linalg.inv(free_poses[k].reshape(4, 4).astype(np.float64))[:3, 3] for k in range(N)])
    cg = h["T_gt"][:N, :3, 3]
    d = np.linalg.norm(ca - cb, axis=1)
    rounds = np.maximum(dbg["rounds"][:, 0], dbg["rounds"][:, 1])
    stats = dict(mode=mode, frames=N, pose_dt_max=float(dt.max()), pose_dt_p99=float(np.percentile(dt, 99)), pose_dr_max=float(dr.max()),
                 frames_within_baseline_tol=float(((dt < TOL_T) & (dr < TOL_R)).mean()),
                 pnp_winner_differs=int(winners_differ), pnp_iterations_differ=int(iters_differ), pnp_inliers_max_diff=int(d_inl.max()),
                 pnp_inliers_frames_differing=int((d_inl > 0).sum()), pnp_pose_dt_max=float(dpnp.max()) if len(dpnp) else 0.0,
                 pnp_pose_dt_p50=float(np.median(dpnp)) if len(dpnp) else 0.0,
                 lm_iterations_max_diff=int(d_it.max()), lm_iterations_frames_differing=int((d_it > 0).sum()),
                 last_map_point_id=int(dbg["new_gid"].max()), max_rounds=int(rounds.max()), frames_over_30_rounds=int((rounds > 30).sum()),
                 resolve_us_median=float(np.median(dbg["resolve_us"][1:])), resolve_us_max=int(dbg["resolve_us"][1:].max()),
                 ate_gpu_vs_free_running_oracle_rmse_m=float(np.sqrt(np.mean(d * d))), ate_gpu_vs_oracle_max_m=float(d.max()),
                 ate_gpu_vs_ground_truth_rmse_m=float(np.sqrt(np.mean(np.sum((ca - cg) ** 2, axis=1)))),
                 ate_oracle_vs_ground_truth_rmse_m=float(np.sqrt(np.mean(np.sum((cb - cg) ** 2, axis=1)))))
    print("FULL_LENGTH " + json.dumps(stats))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(stats, open(os.path.join(out, "full_length_%s.json" % mode), "w"))
        np.savez_compressed(os.path.join(out, "full_length_%s.npz" % mode), dt=dt, dr=dr, d_inl=d_inl, d_it=d_it,
                            pnp_best=dbg["pnp_best"], pnp_iterations=dbg["pnp_iterations"], rounds=dbg["rounds"],
                            resolve_us=dbg["resolve_us"], n_edges=gpu["n_lm_edges"], ate=d)
    if N >= 4541:
        assert stats["last_map_point_id"] > (1 << 20), "the run must take the map-point ids around the position ring"
        assert stats["frames_over_30_rounds"] >= 1, "the run must contain a slow (many-round) frame"
    if mode in ("ord", "exact"):
        # OpenCV's operations and rounding (mode 2, the default: spread over a wavefront; mode 1: one lane, loop by loop): the
        # discrete RANSAC outcome is identical on every frame, poses within BASELINE.md's tolerance
        assert winners_differ == 0 and iters_differ == 0 and d_inl.max() == 0, stats
        assert dt.max() < TOL_T and dr.max() < TOL_R, stats
        assert d_it.max() == 0, stats
        # and FREE-RUNNING (round 6: the LM is order-preserving too): the device's trajectory and the oracle's that never saw a
        # device pose are the same 4,541 x 16 floats, bit for bit - records equal in every field the oracle fills
        first_diff = next((k for k in range(N) if free_poses[k].tobytes() != gpu[k]["Tcw"].tobytes()), None)
        assert first_diff is None, (first_diff, stats)
        assert stats["ate_gpu_vs_free_running_oracle_rmse_m"] == 0.0 and stats["ate_gpu_vs_oracle_max_m"] == 0.0, stats
        assert dt.max() == 0.0 and dr.max() == 0.0, stats
    else:
        # wave-parallel EPnP: the same estimator with another rounding.  A five-point M^T M has a two-dimensional null space
        # whose basis is whatever the eigen-solver's rounding leaves (in OpenCV too); EPnP's N = 1 candidate starts from ONE
        # vector of it, so a sample's pose - and now and then RANSAC's winner - can differ from the sequential order's, and
        # g2o's ten LM iterations with their early-stop rules do not always bring two starts to the same digits.  Measured
        # over the 4,541 frames: identical discrete outcome on 97.3 % of the frames, pose inside BASELINE.md's tolerance on
        # 95.9 %, worst frame 0.19 m, trajectories 6 cm RMSE apart after 4.5 km.  (The index chain is exact in both modes.)
        same = 1.0 - max(winners_differ, iters_differ) / float(N)
        assert same >= 0.965, stats                            # measured: 0.972
        assert stats["frames_within_baseline_tol"] >= 0.95, stats   # measured: 0.959
        assert dt.max() < 0.25 and dr.max() < 6e-4, stats     # measured: 0.186 m, 4.2e-4
        assert d_inl.max() <= 4 and (d_inl > 0).mean() < 0.02, stats   # measured: 3, 1.4 %
        assert d_it.max() <= 1, stats                                   # measured: 1
    # the two free-running trajectories (4.54 km of dead reckoning each) stay together
    assert stats["ate_gpu_vs_free_running_oracle_rmse_m"] < 1.0, stats


@pytest.mark.gpu
def test_front_end_sampled_along_the_headline_sequence(pkg, orc, run):
    """Every 50th of the 4,541 frames (91 stereo pairs, both images): what the batched device front end handed to the tails above
    - keypoints, descriptors, depth of the left image - and the host-buffer entry's full result (right image's keypoints and
    descriptors, uR) against orc_stereo_frame on the re-rendered frame, bit for bit.  The other front-end tests meet the oracle
    on stills and short sequences; this one walks the whole length of the run bench.py times."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    N, h, cam = run["N"], run["host"], run["cam"]
    dev = run["kp"].device
    ks = list(range(0, N, 50))
    frames = {}
    for k in ks:
        L, R, _ = synth.render_sequence(1, seed=synth.BASE_SEED, device=dev, start=k)
        frames[k] = (L[0].cpu().numpy(), R[0].cpu().numpy())
    with ThreadPoolExecutor(8) as ex:                       # the oracle on host threads (ctypes releases the GIL)
        ref = dict(zip(ks, ex.map(lambda k: orc.stereo_frame(frames[k][0], frames[k][1], cam.bf, cam.fx), ks)))
    svo = pkg.Svo(W, H, max_batch=1)
    for k in ks:
        r = ref[k]
        nk = int(h["n"][k])
        assert nk == len(r["kpL"]) > 300, k
        assert h["kp"][k, :nk].tobytes() == r["kpL"].tobytes(), "frame %d: keypoints of the batched front end" % k
        assert np.array_equal(h["desc"][k, :nk], r["dL"]), "frame %d: descriptors of the batched front end" % k
        assert np.array_equal(h["depth"][k, :nk].view(np.uint32), r["depth"].view(np.uint32)), "frame %d: depth" % k
        g = svo.stereo_frame(frames[k][0], frames[k][1], cam)
        for f in ("kpL", "kpR"):
            assert g[f].tobytes() == r[f].tobytes(), (k, f)
        for f in ("dL", "dR"):
            assert np.array_equal(g[f], r[f]), (k, f)
        for f in ("uR", "depth"):
            assert np.array_equal(g[f].view(np.uint32), r[f].view(np.uint32)), (k, f)
    svo.close()
