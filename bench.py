#!/usr/bin/env python3
"""bench.py - stereo frames/s through the MI355X-native tracking path.

The headline is BASELINE.json's metric on its configuration: the FULL Tracking::Track chain
(configs[2]) over a KITTI-00-sized sequence - 4541 DISTINCT synthetic stereo pairs (1241x376
gray, synth-kitti renderer), all resident in HBM before the clock starts.
  --workload track (default): one "step" = svo_track_batch_dev over B consecutive frames of the
        ONE sequence (front end batched over the B pairs, then the ordered tail frame by frame);
        B = 4541 // (warmup + steps), the remainder rides in the first warm-up step so that the
        whole sequence is tracked and ATE is over all of it.
  --workload frontend (BASELINE configs[1], batched): ORB on both images + sparse epipolar
        stereo for B pairs per step, rotating through the resident sequence (every step reads
        B new pairs: the inputs do not stay in the 256 MiB Infinity Cache).
The default line also carries named legs: "frontend", "multi_sequence", "elas", "msa", and the
CPU baseline (the oracle's single-thread port on a bounded prefix, with the GPU-vs-CPU ATE).
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL, used ONLY for the barrier and
the max-over-ranks clock).  The tracking chain of one sequence does not shard (replicas only:
rank r tracks its own sequence, seed + r); the front-end workload shards by stereo pair with no
data-path collective (pair k -> rank k mod N).  Weak scaling.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import svo_loader  # noqa: E402

W, H = 1241, 376
PITCH = 1280                         # HBM row pitch of the resident images (64-byte multiple)
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
S_PYR = sum(1.44 ** -k for k in range(8))   # 3.0957: pyramid pixels / level-0 pixels
ALGO_BYTES_PER_PAIR = 14.0e6         # SURVEY.md section 8d / BASELINE.md section 4
# per-kernel algorithmic bytes per IMAGE (terms of the SURVEY 8d traffic model)
KERNEL_BYTES_PER_IMAGE = {
    "k_fast": S_PYR * W * H,                      # "S [FAST read]"
    "k_pyr_level": (1 + (S_PYR - 1)) * W * H,     # "1 [src read] + (S-1) [pyramid write]"
    "k_pyr_fast": (1 + (S_PYR - 1) + S_PYR) * W * H,
    "k_select": 2 * 500 * 81,                     # Harris: 2N 9x9 windows
    "k_describe": 500 * 37 * 37 + 500 * 60,       # patch gathers + keypoint / descriptor records
}
KERNEL_BYTES_PER_PAIR = {"k_stereo_match": 500 * (11 * 11 + 11 * 21) + 2 * 500 * 32, "k_stereo_median": 500 * 12}


def tail_kernel_bytes(kernel, mean_pool_rows, mean_kp, mean_edges, mean_lm_iters):
    """Algorithmic bytes per FRAME of the three kernels of the ordered tail (SURVEY 8d's
    Hamming and pose-opt terms): what the stage has to touch once, not what the implementation moves."""
    if kernel == "k_ti_lists":      # Hamming: (M + N) descriptors of 32 bytes
        return (mean_pool_rows + mean_kp) * 32.0
    if kernel == "k_ti_resolve":    # packed entries of the M rows in, the compacted pool (descriptor + 10 bytes) out and in
        return mean_pool_rows * (64.0 + 2 * 42.0) + mean_kp * 60.0
    if kernel == "k_tp_hyp":        # the n correspondences (40 bytes each) in, 100 sample records (112 bytes) out
        return mean_edges * 40.0 + 100 * 112.0
    if kernel == "k_tp_frame":      # n correspondences of 40 bytes: once for the gather + once per LM iteration and trial
        return mean_edges * 40.0 * (1.0 + 2.0 * mean_lm_iters)
    return 0.0


def pmc_traffic(kernel, pairs_per_launch=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same
    command (profiles/pmc_latest.json; FETCH_SIZE and WRITE_SIZE collected in separate passes,
    KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))[kernel]
        per_dispatch = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        if pairs_per_launch and d.get("_pairs_per_dispatch_traffic"):    # counters were taken at another batch size: scale
            per_dispatch *= pairs_per_launch / d["_pairs_per_dispatch_traffic"]
        return per_dispatch
    except Exception:
        return None


def pmc_valu(kernel):
    """Integer-VALU view of the same PMC summary (the front-end kernels are VALU-bound):
    instructions per wave and the fraction of SIMD issue time spent on VALU instructions
    (SQ_ACTIVE_INST_VALU counts quad-cycles, 4 per issued wave64 instruction; GRBM_GUI_ACTIVE is summed
    over the 8 XCDs; 1024 SIMDs)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))[kernel]
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        return {"instr_per_wave": d["SQ_INSTS_VALU"] / d["SQ_WAVES"],
                "busy_frac": d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),
                "lds_bank_conflict_frac": d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1.0),
                "source": "profiles/pmc_latest.json"}
    except Exception:
        return None


def render_frames(synth, n, dev, seed, start=0):
    """n consecutive frames resident on `dev`, padded to PITCH; + ground truth.  synth-kitti by
    default; real KITTI 00 frames when KITTI_ROOT is set (layout of the reference's main.cpp:20-57).
    Rendered / loaded in chunks so that only the padded uint8 frames stay resident."""
    root = os.environ.get("KITTI_ROOT")
    dL = torch.zeros((n, H, PITCH), dtype=torch.uint8, device=dev)
    dR = torch.zeros_like(dL)
    Ts = []
    kio = importlib.import_module("stereo_semantic_vo_amd.kitti_io") if root else None
    gt = kio.load_poses(root, "00") if root else None
    for c0 in range(0, n, 64):
        c = min(64, n - c0)
        if root:
            Ln, Rn = kio.load_frames(root, "00", start + c0, c)
            assert Ln.shape[1:] == (H, W), "KITTI 00 frames are expected to be %dx%d" % (W, H)
            T = torch.from_numpy(gt[start + c0:start + c0 + c] if gt is not None else np.tile(np.eye(4), (c, 1, 1)))
            L, R = torch.from_numpy(Ln).to(dev), torch.from_numpy(Rn).to(dev)
        else:
            L, R, T = synth.render_sequence(c, seed=seed, device=dev, start=start + c0)
        dL[c0:c0 + c, :, :W] = L
        dR[c0:c0 + c, :, :W] = R
        Ts.append(T)
    return dL, dR, torch.cat(Ts)


def ate_rmse(res, T_gt):
    """Translation RMSE of the estimated camera centres vs ground truth (no alignment:
    both start at the identity)."""
    err = []
    for k in range(len(res)):
        Tcw = res[k]["Tcw"].reshape(4, 4).astype(np.float64)
        Twc = np.linalg.inv(Tcw)
        err.append(np.linalg.norm(Twc[:3, 3] - T_gt[k][:3, 3]))
    return float(np.sqrt(np.mean(np.square(err)))), float(err[-1])


def cpu_baseline_all_cores(Lh, Rh, cam, budget_s=8.0):
    """Same port, frame-parallel over the host cores (the stateless front end shards by stereo pair
    on the CPU too; ctypes releases the GIL inside the C calls)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as orc
    cores = min(os.cpu_count() or 1, 64)
    deadline = time.perf_counter() + budget_s
    done = [0] * cores

    def worker(t):
        i = t
        while time.perf_counter() < deadline:
            orc.stereo_frame(Lh[i % len(Lh)], Rh[i % len(Lh)], cam.bf, cam.fx)
            done[t] += 1
            i += cores
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": sum(done) / dt, "unit": "stereo pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pairs in %.1f s, %d threads each running the single-thread port on its own pairs"
                      % (sum(done), dt, cores)}


def elas_leg(pkg, W, H, device, d_L, d_R, pitch, B, iters=4):
    """Dense ELAS stereo (BASELINE configs[4] without YOLO): svo_elas_batch_dev on the B
    synthetic pairs already resident in HBM (maps stay in HBM: same boundary as `value`), plus the latency of one
    svo_elas_process call on host buffers.  Beside it the reference's own compiled libelas on one host core
    when oracle/_ref is present."""
    import torch
    ctx = pkg.Svo(W, H, device=device)
    p = pkg.elas_default_params(0)
    dev = d_L.device
    D1 = torch.zeros((B, H, W), dtype=torch.float32, device=dev); D2 = torch.zeros_like(D1)
    torch.cuda.synchronize()
    ctx.elas_batch_dev(d_L.data_ptr(), d_R.data_ptr(), pitch, W, H, B, D1.data_ptr(), D2.data_ptr(), p)
    t0 = time.perf_counter()
    for _ in range(iters):
        produced = ctx.elas_batch_dev(d_L.data_ptr(), d_R.data_ptr(), pitch, W, H, B, D1.data_ptr(), D2.data_ptr(), p)
    thr = B * iters / (time.perf_counter() - t0)
    # per-kernel time per pair (HIP events around every launch on the context's stream) and the HBM roofline of
    # the dominant kernel among those with a defined compulsory traffic (bytes per pair, P = W*H pixels):
    #   k_elas_desc   2P image bytes in, 2*16P descriptor bytes out
    #   k_elas_match  2*16P descriptors + 2*4P owner ids in, 2*4P raw maps out
    #   k_elas_raster 2*4P owner ids written
    #   k_cc_segments 4P map + 3*4P label / run-length / size arrays written and read once
    #   k_elas_gap    2 passes over a 4P map, read + write
    ctx.profile_enable(True); ctx.profile_reset()
    ctx.elas_batch_dev(d_L.data_ptr(), d_R.data_ptr(), pitch, W, H, B, D1.data_ptr(), D2.data_ptr(), p)
    ctx.profile_enable(False)
    P = W * H
    #   k_elas_support 2*16P descriptor bytes read once (HBM-compulsory; the reference's loops re-read 16 KB per lattice
    #                  point and direction - 0.6 GB per pair - which is what the kernel's L1 actually serves)
    #   k_elas_lr / k_elas_mean  two 4P maps read and written
    #   k_elas_planes / k_elas_grid  support-point and triangle lists (tens of KB)
    algo = {"k_elas_desc": 34 * P, "k_elas_match": 48 * P, "k_elas_raster": 8 * P, "k_cc_segments": 28 * P,
            "k_elas_gap": 16 * P, "k_elas_support": 32 * P, "k_elas_lr": 16 * P, "k_elas_mean": 16 * P,
            "k_elas_planes": 2 * 8000 * 36, "k_elas_grid": 2 * (W // 20 + 1) * (H // 20 + 1) * 32 * 2}
    kern = {k: v[0] * 1e3 / B for k, v in ctx.profile().items() if k.startswith("k_")}
    dom = max(kern, key=lambda k: kern[k])
    algo.setdefault(dom, 0)
    ach = algo[dom] / (kern[dom] * 1e-6) / 1e9
    roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": None, "algorithmic_bytes_per_pair": algo[dom],
            "note": "latency / atomic bound, not bandwidth bound: see DESIGN.md section 8"}
    L = d_L[0, :, :W].cpu().numpy(); R = d_R[0, :, :W].cpu().numpy()
    for _ in range(3):
        E1, _ = ctx.elas_process(L, R, p)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.elas_process(L, R, p)
    lat = (time.perf_counter() - t0) / 10
    G1 = D1[0].cpu().numpy()
    ctx.close()
    out = {"value": thr, "unit": "stereo pairs/s", "pairs_per_call": B, "pairs_with_maps": int(produced.sum()),
           "kernel_us_per_pair": {k: round(v, 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])}, "roofline": roof,
           "latency_ms_per_pair_host_buffers": lat * 1e3, "valid_fraction": float((G1 >= 0).mean()),
           "batch_equals_single_call": bool(np.array_equal(G1, E1)), "setting": "ROBOTICS",
           "note": "dense disparity maps D1+D2 per pair; throughput with pairs and maps resident in HBM, latency host to host"}
    try:
        from oracle import binding as ob
        if ob.ref_elas_lib() is not None:
            pb = ob.ref_elas_params(False)
            F1, _ = ob.ref_elas(L, R, pb)
            t0 = time.perf_counter()
            for _ in range(3):
                ob.ref_elas(L, R, pb)
            out["cpu_baseline"] = {"value": 3 / (time.perf_counter() - t0), "unit": "stereo pairs/s", "cores": 1,
                                   "kind": "reference", "sample": "3 calls of the reference's Elas::process on the same pair"}
            out["pixels_differing_from_reference"] = int((F1 != G1).sum())
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline_error"] = str(e)
    return out


def msa_leg(pkg, W, H, device, d_L, d_R):
    """MSA dense stereo (the reference's live frame::MB).  value: throughput of svo_msa_batch_dev over 64
    DISTINCT pairs resident in HBM (maps stay in HBM; the aggregation trees are built on host threads, 16 frames' worth
    at a time, while the previous chunk's level sweeps run) - the same definition as the elas leg.  Beside it: the latency
    of one svo_msa_solve with host buffers, its per-stage GPU times, the tracker with MSA depth, and the CPU restatement
    of the same algorithm on one host core."""
    import torch
    ctx = pkg.Svo(W, H, device=device)
    g2c = lambda g: np.ascontiguousarray(np.repeat(g[:, :, None], 3, 2))
    L = g2c(d_L[0, :, :W].cpu().numpy()); R = g2c(d_R[0, :, :W].cpu().numpy())
    ctx.msa_solve(L, R, 48, 1)
    t0 = time.perf_counter()
    for _ in range(3):
        G = ctx.msa_solve(L, R, 48, 1)
    dt = (time.perf_counter() - t0) / 3
    ctx.profile_enable(True); ctx.profile_reset(); ctx.msa_solve(L, R, 48, 1); ctx.profile_enable(False)
    kern = {k: round(v[0], 3) for k, v in ctx.profile().items() if k.startswith("k_msa")}
    nb = min(64, d_L.shape[0])
    d_disp = torch.zeros((nb, H, W), dtype=torch.float32, device=d_L.device)
    torch.cuda.synchronize()
    batch_rate = 0.0
    for _ in range(2):   # (the first call allocates the chunk arenas)
        t0 = time.perf_counter()
        ctx.msa_batch_dev(d_L.data_ptr(), d_R.data_ptr(), d_L.stride(1), W, H, nb, d_disp.data_ptr(), 48); ctx.sync()
        batch_rate = nb / (time.perf_counter() - t0)
    batch_equals_single = bool(np.array_equal(d_disp[0].cpu().numpy().astype(np.uint8), G))
    del d_disp
    ctx.close()
    # the reference's live configuration end to end: ORB + MSA depth + tracking, 64 frames per call
    cam = pkg.Camera(**pkg.KITTI_00_02)
    trk = pkg.Svo(W, H, max_batch=nb, device=device)
    trk.set_option("depth_source", 2)
    res = torch.zeros((nb, pkg.TRACK_DTYPE.itemsize), dtype=torch.uint8, device=d_L.device)
    torch.cuda.synchronize()
    fps = 0.0
    for _ in range(2):
        trk.track_reset(cam)
        t0 = time.perf_counter()
        trk.track_batch_dev(d_L.data_ptr(), d_R.data_ptr(), d_L.stride(1), nb, res.data_ptr()); trk.sync()
        fps = nb / (time.perf_counter() - t0)
    trk.close()
    out = {"value": batch_rate, "unit": "stereo pairs/s", "pairs_per_call": nb, "batch_equals_single_call": batch_equals_single,
           "latency_ms_per_pair_host_buffers": dt * 1e3, "max_disparity": 48,
           "tracker_frames_per_s_msa_depth_%d_per_call" % nb: fps,
           "gpu_ms_per_pair": kern, "nonzero_fraction": float((G > 0).mean()), "host_cores": os.cpu_count(),
           "note": "d = 48, scale = 1 as frame::MB calls it; gray pair as B = G = R colour images; throughput is bound by the "
                   "host-side tree construction (sequential Chu-Liu/Edmonds + region merging per image, ~0.1 s each, 32 builders side by side)"}
    try:
        from oracle import binding as ob
        ob.build()
        t0 = time.perf_counter()
        F = ob.msa_solve(L, R, 48, 1)
        out["cpu_baseline"] = {"value": 1.0 / (time.perf_counter() - t0), "unit": "stereo pairs/s", "cores": 1, "kind": "port",
                               "sample": "one call of oracle/orc_msa_solve (restatement of MSA::solve) on the same pair"}
        out["pixels_differing_from_port"] = int((F != G).sum())
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline_error"] = str(e)
    return out


def cpu_baseline(Lh, Rh, cam, workload, budget_s=15.0):
    """The oracle (a single-threaded C port of the same path) timed on this box's host
    cores over a bounded sample of the same workload."""
    from oracle import binding as orc
    orc.build()
    n, t0 = 0, time.perf_counter()
    trk = orc.Tracker(W, H, dict(fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, bf=cam.bf)) \
        if workload == "track" else None
    poses = []
    while n < len(Lh) and (time.perf_counter() - t0) < budget_s:
        if trk is not None:
            poses.append(trk.track(Lh[n], Rh[n])[0])
        else:
            orc.stereo_frame(Lh[n], Rh[n], cam.bf, cam.fx)
        n += 1
    dt = time.perf_counter() - t0
    what = "full tracking loop" if workload == "track" else "ORB on L and R + sparse stereo"
    out = {"value": n / dt, "unit": "stereo pairs/s", "cores": 1, "kind": "port",
           "sample": "first %d pairs of the benchmark frames (%s), %.1f s, oracle/libsvo_oracle.so "
                     "single thread; host has %d cores" % (n, what, dt, os.cpu_count() or 0)}
    return out, poses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="track", choices=["track", "frontend"])
    ap.add_argument("--frames", type=int, default=4541, help="length of the resident sequence (KITTI 00: 4541)")
    ap.add_argument("--batch", type=int, default=0,
                    help="pairs per step per GPU: track default frames // (warmup + steps), frontend default 128")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="only the main workload (no frontend / multi_sequence / elas / msa legs)")
    ap.add_argument("--no-elas-leg", action="store_true", help="skip the elas and msa legs")
    ap.add_argument("--no-track-leg", action="store_true", help="(kept for scripts) same as --no-legs for the frontend workload")
    ap.add_argument("--depth-source", type=int, default=0, choices=[0, 1, 2],
                    help="track workload: 0 sparse epipolar stereo (north star), 1 dense ELAS map, 2 dense MSA map")
    ap.add_argument("--sequences", type=int, default=1,
                    help="track workload: S concurrent sequences per GPU (svo_track_multi_step_dev), one frame of each per step")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default) or gloo (single-GPU dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (dry run of the N>1 path on one GPU)")
    args = ap.parse_args()

    pkg = svo_loader.load()
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    shard = importlib.import_module("stereo_semantic_vo_amd.shard")
    rank, world, local = shard.env_rank_world()
    if args.share_gpu:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    cam = pkg.Camera(**pkg.KITTI_00_02)
    track = args.workload == "track"
    multi = track and args.sequences > 1
    nsteps = args.warmup + args.steps
    extra = 0
    if multi:
        B = args.sequences                              # one frame of each of the S sequences per step
        n_frames = B + nsteps
    elif track:
        B = args.batch if args.batch > 0 else max(1, args.frames // nsteps)
        # the remainder of the sequence rides in the first warm-up step, so the whole sequence is tracked
        extra = args.frames - B * nsteps if (args.batch <= 0 and args.warmup > 0 and args.frames > B * nsteps) else 0
        n_frames = B * nsteps + extra
    else:
        B = args.batch if args.batch > 0 else 128
        n_frames = max(args.frames if world == 1 else B * nsteps, 2 * B)
    # the resident sequence: rank r renders its own (tracking: replicas; frontend: pair k -> rank k mod N of ONE sequence)
    seed = shard.sequence_seed_for_rank(synth.BASE_SEED, rank) if track else synth.BASE_SEED
    t_r = time.perf_counter()
    if track or world == 1:
        dL, dR, T_gt = render_frames(synth, n_frames, dev, seed)
    else:
        dL = torch.zeros((n_frames, H, PITCH), dtype=torch.uint8, device=dev)
        dR = torch.zeros_like(dL)
        T_gt = None
        for i, k in enumerate(shard.pairs_for_rank(rank, world, n_frames)):
            fl, fr, _ = render_frames(synth, 1, dev, seed, start=k)
            dL[i] = fl[0]
            dR[i] = fr[0]
    torch.cuda.synchronize()
    render_s = time.perf_counter() - t_r
    frame_bytes = H * PITCH
    rec = pkg.TRACK_DTYPE.itemsize
    svo = pkg.Svo(W, H, device=local, max_kp=500, max_batch=B + extra)
    for kv in filter(None, os.environ.get("SVO_BENCH_OPTIONS", "").split(",")):   # experiments: "track_group=8,..."
        k, v = kv.split("=")
        svo.set_option(k, int(v))
    if multi:
        d_res = torch.zeros((nsteps * B, rec), dtype=torch.uint8, device=dev)
        svo.track_multi_reset(B, cam)
    elif track:
        if args.depth_source:
            svo.set_option("depth_source", args.depth_source)
        d_res = torch.zeros((n_frames, rec), dtype=torch.uint8, device=dev)
        svo.track_reset(cam)
    else:
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        d_depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step(s):
        if multi:
            svo.track_multi_step_dev(dL.data_ptr() + s * frame_bytes, dR.data_ptr() + s * frame_bytes, PITCH, B,
                                     d_res.data_ptr() + s * B * rec)
        elif track:
            off = s * B + (extra if s > 0 else 0)
            nb = B + (extra if s == 0 else 0)
            svo.track_batch_dev(dL.data_ptr() + off * frame_bytes, dR.data_ptr() + off * frame_bytes,
                                PITCH, nb, d_res.data_ptr() + off * rec)
        else:
            off = (s * B) % (n_frames - B + 1)           # every step reads B pairs it has not touched for a long time
            svo.frontend_batch_dev(dL.data_ptr() + off * frame_bytes, dR.data_ptr() + off * frame_bytes, PITCH, B, cam,
                                   d_nL=d_n.data_ptr(), d_depth=d_depth.data_ptr())

    def fence():
        svo.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for s in range(args.warmup):
        step(s)
    fence()
    if not args.no_profile:
        svo.profile_reset()
        svo.profile_enable(True)
    t0 = time.perf_counter()
    for s in range(args.warmup, nsteps):
        step(s)
    fence()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, dist, dev if args.dist_backend == "nccl" else "cpu")
    prof = {}
    if not args.no_profile:
        svo.profile_enable(False)
        prof = svo.profile()

    if rank == 0:
        pairs = world * B * args.steps
        cfg = {"pairs_per_step_per_gpu": B, "resident_pairs_per_gpu": int(n_frames),
               "resident_input_bytes_per_gpu": int(2 * n_frames * frame_bytes), "render_seconds": round(render_s, 1)}
        res = None
        if track:
            res = d_res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
            if multi:                                    # accuracy figures from sequence 0 (starts at frame 0)
                res = res.reshape(-1, B)[:, 0]
                cfg["concurrent_sequences_per_gpu"] = B
            rmse, last = ate_rmse(res, T_gt.numpy()[:len(res)])
            depth_names = {0: "sparse epipolar stereo", 1: "dense ELAS map (svo_elas_batch_dev) -> disp2Depth -> per-keypoint lookups",
                           2: "dense MSA map (svo_msa_batch_dev) -> disp2Depth -> per-keypoint lookups"}
            cfg.update({"depth_source": depth_names[args.depth_source],
                        "workload": "BASELINE configs[2]: synth-kitti00 sequence of %d distinct 1241x376 stereo pairs, full "
                                    "Tracking::Track loop per frame (ORB on L and R, sparse stereo, matching passes 1+2, "
                                    "PnP-RANSAC, pose-only LM, map-point lifecycle); " % len(res) +
                                    ("%d concurrent staggered sequences per GPU, one frame of each per step" % B if multi
                                     else "ONE sequence per GPU in strict frame order (replicas across GPUs)"),
                        "frames_tracked": int(len(res)), "ate_rmse_m_vs_ground_truth": rmse,
                        "final_position_error_m": last, "path_length_m": float(len(res) - 1),
                        "mean_lm_edges": float(res["n_lm_edges"][1:].mean()),
                        "mean_active_rows_pass1_pass2": [float((res["reserved"][1:, 0] & 0xffff).mean()), float((res["reserved"][1:, 1] & 0xffff).mean())],
                        "mean_rounds_pass1_pass2": [float((res["reserved"][1:, 0] >> 16).mean()), float((res["reserved"][1:, 1] >> 16).mean())],
                        "mean_local_map": float(res["n_local_map"][1:].mean()),
                        "tracker_capacity_flag": int(svo.track_overflowed())})
        else:
            n_kp = d_n.cpu().numpy()
            cfg.update({"workload": "BASELINE configs[1], batched: synth-kitti00 stereo pairs 1241x376, ORB pyramid (8 levels, 500 kp) "
                                    "on L and R + sparse epipolar stereo, HBM-resident, every step on %d pairs not touched since "
                                    "%d steps" % (B, max(1, (n_frames - B + 1) // B)),
                        "sharding": "pair k -> rank k mod N, no collective",
                        "mean_keypoints_left": float(n_kp.mean()),
                        "mean_stereo_depths": int((d_depth > 0).sum().item()) / B})
        out = {
            "metric": "stereo frames/sec on KITTI 00 (full Tracking::Track chain)" if track
                      else "stereo frames/sec on KITTI 00 (tracking front end only)",
            "value": pairs / dt, "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "kitti00" if os.environ.get("KITTI_ROOT") else "synthetic", "config": cfg,
        }
        if prof:
            kern = {k: {"avg_ms": v[0] / max(v[1], 1), "launches": v[1], "total_ms": v[0]} for k, v in prof.items()}
            # the kernel the timed region spends most time in.  The tail kernels are timed on every 32nd frame only (an event
            # pair holds the chain up): their totals are avg x launches actually made = one per frame and sequence step
            launches_made = {k: (B * args.steps if track and not multi and k.startswith(("k_ti_", "k_tp_")) else v[1]) for k, v in prof.items()}
            for k in kern:
                kern[k]["launches_in_timed_region"] = launches_made[k]
                kern[k]["total_ms_estimated"] = kern[k]["avg_ms"] * launches_made[k]
            dom = max(kern.items(), key=lambda kv: kv[1]["total_ms_estimated"])[0]
            dom_s = prof[dom][0] / max(prof[dom][1], 1) * 1e-3
            units = "images"
            if dom in KERNEL_BYTES_PER_IMAGE:
                algo = KERNEL_BYTES_PER_IMAGE[dom] * 2 * B
            elif dom in KERNEL_BYTES_PER_PAIR:
                algo = KERNEL_BYTES_PER_PAIR[dom] * B
            else:
                units = "frames"
                mp = float(res["n_local_map"][1:].mean()) + 500.0 if res is not None else 0.0
                algo = tail_kernel_bytes(dom, mp, float(res["n_kp"].mean()), float(res["n_lm_edges"][1:].mean()),
                                         float(res["lm_iterations"][1:].mean())) * (B if multi else 1)
            ach = algo / dom_s / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": pmc_traffic(dom),
                               "algorithmic_bytes_per_launch": algo, "launch_covers": "%s of one launch" % units,
                               "share_of_timed_kernel_time": kern[dom]["total_ms_estimated"] / max(sum(v["total_ms_estimated"] for v in kern.values()), 1e-9),
                               "valu": pmc_valu(dom),
                               "pipeline_frac": ALGO_BYTES_PER_PAIR * (pairs / dt / world) / 1e9 / HBM_PEAK_GBS,
                               "note": "the ordered tail is a dependent chain of single-workgroup kernels: latency-bound, far "
                                       "from the HBM roof by construction (DESIGN.md section 5)" if units == "frames" else None}
            out["kernels"] = kern
        if world == 1 and not args.no_cpu_baseline:
            ns = min(n_frames, 64)
            Lh = dL[:ns, :, :W].cpu().numpy()
            Rh = dR[:ns, :, :W].cpu().numpy()
            out["cpu_baseline"], cpu_poses = cpu_baseline(Lh, Rh, cam, args.workload)
            if track and not multi and cpu_poses:
                # ATE of the GPU trajectory against the CPU port's on the same frames (BASELINE metric: "ATE vs CPU reference")
                k = len(cpu_poses)
                ca = np.array([np.linalg.inv(res[i]["Tcw"].reshape(4, 4).astype(np.float64))[:3, 3] for i in range(k)])
                cb = np.array([np.linalg.inv(cpu_poses[i]["Tcw"].reshape(4, 4).astype(np.float64))[:3, 3] for i in range(k)])
                d = np.linalg.norm(ca - cb, axis=1)
                out["ate_vs_cpu_m"] = {"rmse": float(np.sqrt(np.mean(d * d))), "max": float(d.max()), "frames": int(k)}
                out["cpu_baseline"]["counters_identical_to_gpu"] = bool(all(
                    all(int(res[i][f]) == int(cpu_poses[i][f]) for f in ("n_kp", "n_stereo", "n_match_pass1", "n_match_pass2",
                                                                       "n_lm_edges", "n_new_mappoints", "n_local_map"))
                    for i in range(k)))
        legs = world == 1 and not args.no_legs and not multi and args.depth_source == 0
        if legs and track:
            svo.profile_enable(False)
            out["frontend"] = frontend_leg(pkg, svo, cam, dL, dR, n_frames, frame_bytes, dev,
                                           None if args.no_cpu_baseline else cpu_baseline_all_cores)
            out["multi_sequence"] = multi_sequence_leg(pkg, svo, cam, dL, dR, frame_bytes, rec, dev, res)
        if legs and not args.no_elas_leg:
            out["elas"] = elas_leg(pkg, W, H, dev.index or 0, dL, dR, PITCH, min(512, int(n_frames)), iters=2)
            out["msa"] = msa_leg(pkg, W, H, dev.index or 0, dL, dR)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    svo.close()


def frontend_leg(pkg, svo, cam, dL, dR, n_frames, frame_bytes, dev, all_cores):
    """BASELINE configs[1], batched: ORB on both images + sparse stereo, 384 pairs per step, every step on pairs the
    GPU has not touched for thousands of frames (the sequence is far larger than the Infinity Cache)."""
    B, steps = int(os.environ.get("SVO_BENCH_FE_B", "384")), 16
    fe = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=B)
    for kv in filter(None, os.environ.get("SVO_BENCH_FE_OPTIONS", "").split(",")):   # experiments: "frontend_overlap=4"
        k, v = kv.split("=")
        fe.set_option(k, int(v))
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)

    def run(s):
        off = (s * B) % (n_frames - B + 1)
        fe.frontend_batch_dev(dL.data_ptr() + off * frame_bytes, dR.data_ptr() + off * frame_bytes, PITCH, B, cam,
                              d_nL=d_n.data_ptr(), d_depth=d_depth.data_ptr())
    for s in range(3):
        run(s)
    fe.sync()
    # throughput: the library's normal mode (the two halves of a batch side by side on two streams) ...
    t0 = time.perf_counter()
    for s in range(3, 3 + steps):
        run(s)
    fe.sync()
    dt = time.perf_counter() - t0
    # ... per-kernel times: a second pass over other pairs with the timers on - one chain on one stream, each kernel alone
    fe.profile_reset(); fe.profile_enable(True)
    t0 = time.perf_counter()
    for s in range(3 + steps, 3 + 2 * steps):
        run(s)
    fe.sync()
    dt_single = time.perf_counter() - t0
    fe.profile_enable(False)
    prof = fe.profile()
    fe.close()
    kern = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}
    dom = max(prof.items(), key=lambda kv: kv[1][0])[0]
    algo = KERNEL_BYTES_PER_IMAGE.get(dom, 0) * 2 * B if dom in KERNEL_BYTES_PER_IMAGE else KERNEL_BYTES_PER_PAIR.get(dom, 0) * B
    ach = algo / (prof[dom][0] / max(prof[dom][1], 1) * 1e-3) / 1e9
    out = {"value": B * steps / dt, "unit": "stereo pairs/s", "pairs_per_step": B, "steps": steps,
           "distinct_input_bytes_read": int(2 * B * steps * frame_bytes), "kernel_avg_ms": kern,
           "value_one_stream_with_timers": B * steps / dt_single,
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, B), "valu": pmc_valu(dom),
                        "algorithmic_bytes_per_launch": algo,
                        "pipeline_frac": ALGO_BYTES_PER_PAIR * (B * steps / dt) / 1e9 / HBM_PEAK_GBS}}
    if all_cores is not None:
        Lh = dL[:64, :, :W].cpu().numpy(); Rh = dR[:64, :, :W].cpu().numpy()
        out["cpu_baseline_all_cores"] = all_cores(Lh, Rh, cam)
    return out


def multi_sequence_leg(pkg, svo, cam, dL, dR, frame_bytes, rec, dev, single):
    """S staggered sequences advanced together (one workgroup per sequence in every tail kernel); sequence 0 must
    reproduce the single chain's records."""
    S, msteps = 64, 48
    ms = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=S)
    ms.set_option("multi_pipeline", int(os.environ.get("SVO_BENCH_MULTI_PIPELINE", "1")))   # front end of step t + 1 beside the tail of step t
    mres = torch.zeros((msteps * S, rec), dtype=torch.uint8, device=dev)
    ms.track_multi_reset(S, cam)
    for t in range(2):
        ms.track_multi_step_dev(dL.data_ptr() + t * frame_bytes, dR.data_ptr() + t * frame_bytes, PITCH, S,
                                mres.data_ptr() + t * S * rec)
    ms.track_multi_reset(S, cam)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for t in range(msteps):
        ms.track_multi_step_dev(dL.data_ptr() + t * frame_bytes, dR.data_ptr() + t * frame_bytes, PITCH, S,
                                mres.data_ptr() + t * S * rec)
    ms.sync()
    mdt = time.perf_counter() - t1
    m = mres.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(msteps, S)
    ms.close()
    same = all(m[t, 0].tobytes() == single[t].tobytes() for t in range(msteps))
    return {"value": msteps * S / mdt, "unit": "stereo frames/s", "sequences": S, "steps": msteps,
            "sequence0_equals_single_chain": bool(same),
            "pipelined_steps": bool(int(os.environ.get("SVO_BENCH_MULTI_PIPELINE", "1"))),
            "note": "SURVEY 8e's 'G independent sequences' variant on ONE GPU: full Tracking::Track per frame; with "
                    "pipelined_steps the stateless front end of step t + 1 runs beside the tail of step t (multi_pipeline option)"}


if __name__ == "__main__":
    main()
