#!/usr/bin/env python3
"""bench.py - stereo frames/s through the MI355X-native tracking path.

The headline is BASELINE.json's metric on its configuration: the FULL Tracking::Track chain
(configs[2]) over a KITTI-00-sized sequence - 4541 DISTINCT synthetic stereo pairs (1241x376
gray, synth-kitti renderer), all resident in HBM before the clock starts.
  --workload track (default): one "step" = svo_track_batch_dev over B consecutive frames of the
        ONE sequence (front end batched over the B pairs, then the ordered tail frame by frame);
        B = 4541 // (warmup + steps), the remainder rides in the first warm-up step so that the
        whole sequence is tracked and ATE is over all of it.  The timed region runs the product's
        default schedule (no profiling hooks); per-kernel HIP-event times come from a separate,
        untimed pass over the first frames, the tail's critical path from in-kernel wall-clock
        stamps of the timed region itself.
  --workload frontend (BASELINE configs[1], batched): ORB on both images + sparse epipolar
        stereo for B pairs per step, rotating through the resident sequence.
  --shard: BASELINE configs[3] as the reference would run it - ONE sequence, ONE process, --gpus G
        contexts (one per GPU when the box has G GPUs, otherwise all on GPU 0): pair k's front end
        on context k mod G, the ordered tail on context 0 (svo_track_sharded_dev).
The default line also carries named legs: "frontend", "multi_sequence", "sharded", "semantic_elas"
(configs[4] as a whole), "elas", "msa", and the CPU baseline (the oracle's single-thread port on a
bounded prefix; its tail alone over ALL tracked frames for the counter / ATE comparison).
N > 1: `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run as a child
process, before this process touches a GPU) unless it already runs under one; one process per GPU
(backend nccl = RCCL, used ONLY for the barrier and the max-over-ranks clock).  The tracking chain
of one sequence does not shard (replicas: rank r tracks its own sequence, seed + r); the front-end
workload shards by stereo pair with no data-path collective (pair k -> rank k mod N).  Weak
scaling.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

T_START = time.perf_counter()

# (Up to round 4 this file raised GPU_MAX_HW_QUEUES to 16 so that a context's streams would not share a hardware queue.  The
# library now picks streams that run side by side by measuring - svo_pick_stream, tests/test_shard.py
# test_tail_rate_does_not_depend_on_the_process_stream_history - and the bench runs with the runtime's default, like a deployment.)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1241, 376
PITCH = 1280                         # HBM row pitch of the resident images (64-byte multiple)
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
S_PYR = sum(1.44 ** -k for k in range(8))   # 3.0957: pyramid pixels / level-0 pixels
ALGO_BYTES_PER_PAIR = 14.0e6         # SURVEY.md section 8d / BASELINE.md section 4
# per-kernel algorithmic bytes per IMAGE (terms of the SURVEY 8d traffic model)
KERNEL_BYTES_PER_IMAGE = {
    "k_fast": S_PYR * W * H,                      # "S [FAST read]"
    "k_pyr_level": (1 + (S_PYR - 1)) * W * H,     # "1 [src read] + (S-1) [pyramid write]" - the pyramid's launches together
    "k_pyr_fused": (1 + (S_PYR - 1)) * W * H,     #   (three fused launches by default, seven with pyr_fused = 0)
    "k_select": 2 * 500 * 81,                     # Harris: 2N 9x9 windows
    "k_describe": 500 * 37 * 37 + 500 * 60,       # patch gathers + keypoint / descriptor records
}
KERNEL_BYTES_PER_PAIR = {"k_stereo_match": 500 * (11 * 11 + 11 * 21) + 2 * 500 * 32, "k_stereo_median": 500 * 12}


def host_cpus():
    """CPUs this process may actually use: the machine's, cut down to the cgroup quota (the GPU boxes give a process
    16 CPUs of a 256-thread machine)."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, round(int(q) / int(p))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def tail_kernel_bytes(kernel, mean_pool_rows, mean_kp, mean_edges, mean_lm_iters):
    """Algorithmic bytes per FRAME of the kernels of the ordered tail (SURVEY 8d's
    Hamming and pose-opt terms): what the stage has to touch once, not what the implementation moves."""
    if kernel == "k_ti_lists":      # Hamming: (M + N) descriptors of 32 bytes
        return (mean_pool_rows + mean_kp) * 32.0
    if kernel == "k_ti_resolve":    # packed entries of the M rows in, the compacted pool (descriptor + 10 bytes) out and in
        return mean_pool_rows * (64.0 + 2 * 42.0) + mean_kp * 60.0
    if kernel in ("k_tp_hyp", "k_tp_hyp_exact", "k_tp_hyp_ord"):   # the n correspondences (40 bytes each) in, 100 sample records (112 bytes) out
        return mean_edges * 40.0 + 100 * 112.0
    if kernel == "k_tp_frame":      # n correspondences of 40 bytes: once for the gather + once per LM iteration and trial
        return mean_edges * 40.0 * (1.0 + 2.0 * mean_lm_iters)
    if kernel in ("k_tp_pose", "k_tp_tail_ord"):       # RANSAC samples + LM in one launch: both of the above
        return mean_edges * 40.0 * (2.0 + 2.0 * mean_lm_iters) + 100 * 112.0
    return 0.0


def pmc_entry(kernel):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))[kernel]
    except Exception:
        return None


def pmc_traffic(kernel, pairs_per_launch=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same
    command (profiles/pmc_latest.json; FETCH_SIZE and WRITE_SIZE collected in separate passes,
    KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    d = pmc_entry(kernel)
    try:
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
    d = pmc_entry(kernel)
    try:
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        return {"instr_per_wave": d["SQ_INSTS_VALU"] / d["SQ_WAVES"],
                # SQ_WAVE_CYCLES counts quad-cycles a wave is resident: its issue opportunities (one instruction per quad-cycle at most)
                "inst_per_quad_cycle_per_wave": (d["SQ_INSTS_VALU"] + d.get("SQ_INSTS_SALU", 0.0) + d.get("SQ_INSTS_LDS", 0.0)) / d["SQ_WAVE_CYCLES"],
                "instr_all_per_wave": (d["SQ_INSTS_VALU"] + d.get("SQ_INSTS_SALU", 0.0) + d.get("SQ_INSTS_LDS", 0.0)) / d["SQ_WAVES"],
                "busy_frac": d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),
                "lds_bank_conflict_frac": d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1.0),
                "source": "profiles/pmc_latest.json"}
    except Exception:
        return None


def render_frames(synth, n, dev, seed, start=0):
    """n consecutive frames resident on `dev`, padded to PITCH; + ground truth.  synth-kitti by
    default; real KITTI 00 frames when KITTI_ROOT is set (layout of the reference's main.cpp:20-57).
    Rendered / loaded in chunks so that only the padded uint8 frames stay resident."""
    import numpy as np
    import torch
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


def centres(poses):
    import numpy as np
    return np.array([np.linalg.inv(p.reshape(4, 4).astype(np.float64))[:3, 3] for p in poses])


def ate_rmse(res, T_gt):
    """Translation RMSE of the estimated camera centres vs ground truth (no alignment:
    both start at the identity)."""
    import numpy as np
    err = np.linalg.norm(centres(res["Tcw"]) - T_gt[:len(res), :3, 3], axis=1)
    return float(np.sqrt(np.mean(np.square(err)))), float(err[-1])


def moving_boxes(k):
    """Offline detections of frame k in the reference's {left, right, top, bottom} layout (main.cpp:82-95): one box
    crossing the image at 4 px per frame (a vehicle), one fixed box growing slowly (something approached)."""
    x = 80 + (4 * k) % 820
    return [[x, x + 260, 150, 330], [100, 260, 200, 300 + (2 * k) % 60]]


def boxes_hbm(pkg, n, dev):
    import numpy as np
    import torch
    b = np.zeros((n, 2, 4), np.int32)
    for k in range(n):
        b[k] = moving_boxes(k)
    tb = torch.from_numpy(b).to(dev)
    tn = torch.full((n,), 2, dtype=torch.int32, device=dev)
    return pkg.boxes_dev(tb.data_ptr(), tn.data_ptr(), 2), (tb, tn)


def cpu_baseline_all_cores(Lh, Rh, cam, budget_s=8.0):
    """Same port, frame-parallel over the host cores this process may use (the stateless front end shards by stereo
    pair on the CPU too; ctypes releases the GIL inside the C calls)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as orc
    cores = min(host_cpus(), 64)
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
            "sample": "%d pairs in %.1f s, %d threads (= the CPUs this process may use: cgroup quota; the machine shows %d) "
                      "each running the single-thread port on its own pairs" % (sum(done), dt, cores, os.cpu_count() or 0)}


def cpu_baseline(Lh, Rh, cam, workload, budget_s=15.0):
    """The oracle (a single-threaded C port of the same path) timed on this box's host
    cores over a bounded sample of the same workload."""
    from oracle import binding as orc
    orc.build()
    n, t0 = 0, time.perf_counter()
    trk = orc.Tracker(W, H, dict(fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, bf=cam.bf)) \
        if workload == "track" else None
    while n < len(Lh) and (time.perf_counter() - t0) < budget_s:
        if trk is not None:
            trk.track(Lh[n], Rh[n])
        else:
            orc.stereo_frame(Lh[n], Rh[n], cam.bf, cam.fx)
        n += 1
    dt = time.perf_counter() - t0
    what = "full tracking loop" if workload == "track" else "ORB on L and R + sparse stereo"
    return {"value": n / dt, "unit": "stereo pairs/s", "cores": 1, "kind": "port",
            "sample": "first %d pairs of the benchmark frames (%s), %.1f s, oracle/libsvo_oracle.so "
                      "single thread; this process may use %d CPUs (cgroup quota) of the machine's %d"
                      % (n, what, dt, host_cpus(), os.cpu_count() or 0)}


ORACLE_TAIL_CHILD = r"""
import sys, time
import numpy as np
root, inp, out = sys.argv[1:4]
sys.path.insert(0, root)
from oracle import binding as orc
orc.build()
d = np.load(inp)
W, H = (int(v) for v in d["wh"])
fx, fy, cx, cy, bf = (float(v) for v in d["cam"])
kp, desc, n, depth = d["kp"], d["desc"], d["n"], d["depth"]
t0 = time.perf_counter()
trk = orc.Tracker(W, H, dict(fx=fx, fy=fy, cx=cx, cy=cy, bf=bf))
recs = []
for k in range(len(n)):
    nk = int(n[k])
    recs.append(trk.track_tail(kp[k, :nk], desc[k, :nk], depth[k, :nk])[0].copy())
trk.close()
np.savez(out, records=np.array(recs), seconds=time.perf_counter() - t0)
"""


class OracleTailRun:
    """The oracle's ordered tail (orc_track_tail) over ALL tracked frames, fed with the device front end's outputs (bit-exact
    against the oracle's own front end by the parity tests), free-running in a CHILD PROCESS while the GPU legs run (one of
    the host's cores; as a thread of this process it held the interpreter lock often enough to halve the host-bound legs):
    every counter of every frame against the device's records, and the ATE between the two whole trajectories.  The child
    only loads numpy and the oracle library - it never touches the GPU."""

    def __init__(self, cam, kp, desc, n, depth):
        import tempfile
        import numpy as np
        self.records, self.seconds, self.error, self.proc = [], 0.0, None, None
        self.dir = tempfile.mkdtemp(prefix="svo_oracle_tail_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        self.inp, self.out = os.path.join(self.dir, "in.npz"), os.path.join(self.dir, "out.npz")
        np.savez(self.inp, kp=kp, desc=desc, n=n, depth=depth, wh=np.array([W, H]), cam=np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.bf]))

    def start(self):
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", ORACLE_TAIL_CHILD, ROOT, self.inp, self.out],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        except Exception as e:  # noqa: BLE001
            self.error = repr(e)

    def join(self):
        import shutil
        import numpy as np
        try:
            if self.proc is not None:
                _, err = self.proc.communicate(timeout=1800)
                if self.proc.returncode != 0:
                    self.error = "oracle tail child failed: " + (err or "")[-500:]
                else:
                    d = np.load(self.out)
                    self.records, self.seconds = list(d["records"]), float(d["seconds"])
        except Exception as e:  # noqa: BLE001
            self.error = repr(e)
        finally:
            shutil.rmtree(self.dir, ignore_errors=True)

    def report(self, res):
        import numpy as np
        if self.error or not self.records:
            return {"error": self.error or "no frames"}
        k = len(self.records)
        cnt = ("n_kp", "n_stereo", "n_match_pass1", "n_match_pass2", "n_lm_edges", "n_new_mappoints", "n_local_map")
        same = all(all(int(res[i][f]) == int(self.records[i][f]) for f in cnt) for i in range(k))
        d = np.linalg.norm(centres(res["Tcw"][:k]) - centres([r["Tcw"] for r in self.records]), axis=1)
        return {"frames": k, "counters_identical_to_gpu": bool(same), "seconds": round(self.seconds, 1),
                "ate_vs_cpu_m": {"rmse": float(np.sqrt(np.mean(d * d))), "max": float(d.max()), "frames": int(k)},
                "what": "oracle/orc_track_tail (free-running, single thread) on the device front end's keypoints / descriptors / depths"}


def frontend_outputs(pkg, cam, dL, dR, n_frames, dev):
    """Keypoints, descriptors, counts and depths of all resident frames (device front end), on the host."""
    import torch
    K = 500
    kp = torch.zeros((n_frames, K, pkg.KP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    desc = torch.zeros((n_frames, K, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(n_frames, dtype=torch.int32, device=dev)
    depth = torch.zeros((n_frames, K), dtype=torch.float32, device=dev)
    fe = pkg.Svo(W, H, device=dev.index or 0, max_kp=K, max_batch=128, flags=pkg.CREATE_TAIL_ALL_CUS)
    fb = H * PITCH
    for c0 in range(0, n_frames, 128):
        c = min(128, n_frames - c0)
        fe.frontend_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, PITCH, c, cam, d_kpL=kp[c0:].data_ptr(),
                              d_descL=desc[c0:].data_ptr(), d_nL=n[c0:].data_ptr(), d_depth=depth[c0:].data_ptr())
    fe.sync()
    fe.close()
    return (kp.cpu().numpy().view(pkg.KP_DTYPE).reshape(n_frames, K), desc.cpu().numpy(), n.cpu().numpy(), depth.cpu().numpy())


# ------------------------------------------------------------------------------------------------ legs
def elas_leg(pkg, device, d_L, d_R, pitch, B, iters=4):
    """Dense ELAS stereo: svo_elas_batch_dev on the B synthetic pairs already resident in HBM (maps
    stay in HBM: same boundary as `value`), plus the latency of one svo_elas_process call on host buffers.  Beside it
    the reference's own compiled libelas on one host core when oracle/_ref is present."""
    import numpy as np
    import torch
    # (a context that never runs the batched tracker: its first queue - where every stage of these entries runs - keeps all CUs)
    ctx = pkg.Svo(W, H, device=device, flags=pkg.CREATE_TAIL_ALL_CUS)
    p = pkg.elas_default_params(0)
    dev = d_L.device
    D1 = torch.zeros((B, H, W), dtype=torch.float32, device=dev); D2 = torch.zeros_like(D1)
    torch.cuda.synchronize()
    ctx.elas_batch_dev(d_L.data_ptr(), d_R.data_ptr(), pitch, W, H, B, D1.data_ptr(), D2.data_ptr(), p)
    t0 = time.perf_counter()
    for _ in range(iters):
        produced = ctx.elas_batch_dev(d_L.data_ptr(), d_R.data_ptr(), pitch, W, H, B, D1.data_ptr(), D2.data_ptr(), p)
    thr = B * iters / (time.perf_counter() - t0)
    ctx.profile_enable(True); ctx.profile_reset()
    ctx.elas_batch_dev(d_L.data_ptr(), d_R.data_ptr(), pitch, W, H, B, D1.data_ptr(), D2.data_ptr(), p)
    ctx.profile_enable(False)
    P = W * H
    # compulsory traffic per pair (bytes, P = W*H pixels):
    #   k_elas_desc 2P image bytes in, 2*16P descriptor bytes out; k_elas_match 2*16P descriptors + 2*4P owner ids in,
    #   2*4P raw maps out; k_elas_raster 2*4P owner ids; k_cc_segments 4P map + 3*4P label arrays; k_elas_gap 2 passes over
    #   a 4P map; k_elas_support 2*16P descriptors read once; k_elas_lr / k_elas_mean two 4P maps read and written
    algo = {"k_elas_desc": 34 * P, "k_elas_match": 48 * P, "k_elas_raster": 8 * P, "k_cc_segments": 28 * P,
            "k_elas_gap": 16 * P, "k_elas_support": 32 * P, "k_elas_lr": 16 * P, "k_elas_mean": 16 * P,
            "k_elas_planes": 2 * 8000 * 36, "k_elas_grid": 2 * (W // 20 + 1) * (H // 20 + 1) * 32 * 2}
    kern = {k: v[0] * 1e3 / B for k, v in ctx.profile().items() if k.startswith("k_")}
    dom = max(kern, key=lambda k: kern[k])
    algo.setdefault(dom, 0)
    ach = algo[dom] / (kern[dom] * 1e-6) / 1e9
    roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic(dom, 1), "algorithmic_bytes_per_pair": algo[dom],
            "pipeline_frac": 198.0 * P * thr / 1e9 / HBM_PEAK_GBS,
            "note": "latency / atomic bound, not bandwidth bound: see DESIGN.md section 8; pipeline_frac = sum of the stages' "
                    "compulsory bytes (198 P per pair) x pairs/s over the HBM peak; the batch is bound by the host-side support-point "
                    "filter + Delaunay triangulation under the box's CPU quota"}
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
           "batch_equals_single_call": bool(np.array_equal(G1, E1)), "setting": "ROBOTICS", "host_cpus": host_cpus(),
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


def msa_leg(pkg, device, d_L, d_R):
    """MSA dense stereo (the reference's live frame::MB).  value: throughput of svo_msa_batch_dev over 64
    DISTINCT pairs resident in HBM (maps stay in HBM; the aggregation trees are built on host threads, 16 frames' worth
    at a time, while the previous chunk's level sweeps run) - the same definition as the elas leg.  Beside it: the latency
    of one svo_msa_solve with host buffers, its per-stage GPU times, the tracker with MSA depth, and the CPU restatement
    of the same algorithm on one host core."""
    import numpy as np
    import torch
    ctx = pkg.Svo(W, H, device=device, flags=pkg.CREATE_TAIL_ALL_CUS)
    g2c = lambda g: np.ascontiguousarray(np.repeat(g[:, :, None], 3, 2))
    L = g2c(d_L[0, :, :W].cpu().numpy()); R = g2c(d_R[0, :, :W].cpu().numpy())
    ctx.msa_solve(L, R, 48, 1)
    t0 = time.perf_counter()
    for _ in range(3):
        G = ctx.msa_solve(L, R, 48, 1)
    dt = (time.perf_counter() - t0) / 3
    ctx.profile_enable(True); ctx.profile_reset(); ctx.msa_solve(L, R, 48, 1); ctx.profile_enable(False)
    prof = ctx.profile()
    kern = {k: round(v[0], 3) for k, v in prof.items() if k.startswith("k_msa")}
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
    # roofline of the dominant GPU kernel: the tree aggregation (MSA::TreeDp, 3 aggregations per solve, each two sweeps over
    # a P x 49 float cost volume: read cost, write costUp, read both + parents, write costA ~ 3 volumes read + 2 written)
    P, D = W * H, 49
    dom = max(kern, key=lambda k: kern[k]) if kern else None
    algo = 3 * 5 * P * D * 4.0 if dom and "dp" in dom else 2 * P * D * 4.0
    roof = None
    if dom:
        ach = algo / (kern[dom] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                # (the timer around an aggregation is named k_msa_tree_dp; the kernel rocprofv3 sees inside it is k_msa_dp_bfs)
                "traffic": (pmc_traffic({"k_msa_tree_dp": "k_msa_dp_bfs"}.get(dom, dom)) or 0) * prof[dom][1] or None, "algorithmic_bytes_per_solve": algo, "launches_per_solve": prof[dom][1],
                "note": "one workgroup per disparity walks ~2000 barrier-separated tree levels: latency bound; the solve as a whole is "
                        "bound by the host-side tree construction (see `note`)"}
    out = {"value": batch_rate, "unit": "stereo pairs/s", "pairs_per_call": nb, "batch_equals_single_call": batch_equals_single,
           "latency_ms_per_pair_host_buffers": dt * 1e3, "max_disparity": 48,
           "tracker_frames_per_s_msa_depth_%d_per_call" % nb: fps, "roofline": roof,
           "gpu_ms_per_pair": kern, "nonzero_fraction": float((G > 0).mean()), "host_cpus": host_cpus(),
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


def frontend_leg(pkg, cam, dL, dR, n_frames, frame_bytes, dev, all_cores):
    """BASELINE configs[1], batched: ORB on both images + sparse stereo, 384 pairs per step, every step on pairs the
    GPU has not touched for thousands of frames (the sequence is far larger than the Infinity Cache)."""
    import torch
    B, steps = int(os.environ.get("SVO_BENCH_FE_B", "384")), 16
    fe = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=B, flags=pkg.CREATE_TAIL_ALL_CUS)   # front end only: all CUs for slice 0's queue
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
    # throughput: the library's normal mode (slices of a batch side by side on their own streams) ...
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
    # What bounds the front end: vector-ALU instruction issue, priced PER ISSUE CLASS.  A SIMD does not issue "one wave64
    # instruction per four cycles" (round 4's figure - the cadence of ONE wave): tools/microbench/valu_class (profiles/
    # r05_valu_class.jsonl, 1 / 2 / 4 / 8 waves per SIMD) measures ~1.0 ns per instruction for plain VOP1 / VOP2 integer and f32
    # operations and ~1.85 ns for everything else (three-operand VOP3, packed, min / max, compares, multiplies, dots, v_perm,
    # SDWA / DPP, f64, and any VOP2 that reads a scalar register).  Each kernel's counted instructions (SQ_INSTS_VALU of the
    # committed PMC passes, 64 pairs per dispatch) are weighted with the class mix of its emitted code (tools/isa_mix.py ->
    # profiles/r05_isa_mix.json): the time the chip's 1,024 SIMDs need for them is the bound.
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "r05_isa_mix.json")))
        tf, th, tq = mix["t_full_ns"], mix["t_half_ns"], mix["t_quarter_ns"]
        per_kernel, ns_total, instr_total = {}, 0.0, 0.0
        for k, launches, variants in (("k_pyr_fused", 3, ("k_pyr_fused<2>", "k_pyr_fused<3>")), ("k_fast", 1, ("k_fast",)), ("k_select", 1, ("k_select",)),
                                      ("k_describe", 1, ("k_describe",)), ("k_stereo_match", 1, ("k_stereo_match",)),
                                      ("k_stereo_median", 1, ("k_stereo_median",))):
            e = pmc_entry(k)
            n_instr = launches * e["SQ_INSTS_VALU"] / (e.get("_pairs_per_dispatch_traffic", 64) / 64.0) / 64.0     # per pair
            m = [mix["kernels"][v] for v in variants]
            ns = sum(x["share_full"] * tf + x["share_half"] * th + x["share_quarter"] * tq for x in m) / len(m)
            per_kernel[k] = {"valu_instructions_per_pair": n_instr, "share_half_rate": sum(x["share_half"] for x in m) / len(m), "ns_per_instruction": ns}
            ns_total += n_instr * ns
            instr_total += n_instr
        simds = 256 * 4
        bound_pairs_per_s = simds / (ns_total * 1e-9)
        out["roofline"]["valu_issue"] = {
            "bound": "wave64 VALU instruction issue, weighted by issue class (full-rate %.2f ns, half-rate %.2f ns, quarter-rate %.2f ns per "
                     "instruction per SIMD; 1,024 SIMDs)" % (tf, th, tq),
            "pairs_per_s_at_the_bound": bound_pairs_per_s, "achieved_pairs_per_s": out["value"], "frac": out["value"] / bound_pairs_per_s,
            "valu_instructions_per_pair": instr_total, "simd_ns_per_pair": ns_total, "kernels": per_kernel,
            "frac_if_every_instruction_were_full_rate": out["value"] * instr_total * tf * 1e-9 / simds,
            "frac_if_every_instruction_were_half_rate": out["value"] * instr_total * th * 1e-9 / simds,
            "source": "profiles/pmc_latest.json (SQ_INSTS_VALU), profiles/r05_isa_mix.json (static class mix of the emitted code), "
                      "profiles/r05_valu_class.jsonl (measured issue cost per instruction)"}
    except Exception as e:  # noqa: BLE001
        out["roofline"]["valu_issue"] = {"error": repr(e)}
    if all_cores is not None:
        Lh = dL[:64, :, :W].cpu().numpy(); Rh = dR[:64, :, :W].cpu().numpy()
        out["cpu_baseline_all_cores"] = all_cores(Lh, Rh, cam)
    return out


def multi_sequence_leg(pkg, cam, dL, dR, frame_bytes, rec, dev, single):
    """S staggered sequences advanced together (one workgroup per sequence in every tail kernel); sequence 0 must
    reproduce the single chain's records."""
    import torch
    S, msteps = int(os.environ.get("SVO_BENCH_MULTI_S", "64")), 48   # (the leg is quoted at 64 sequences; the switch is for tools/option sweeps)
    ms = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=S)
    ms.set_option("multi_pipeline", int(os.environ.get("SVO_BENCH_MULTI_PIPELINE", "1")))   # front end of step t + 1 beside the tail of step t
    if os.environ.get("SVO_BENCH_TAIL_SEMI"):
        ms.set_option("tail_semi", int(os.environ["SVO_BENCH_TAIL_SEMI"]))   # 2: the steps' pose chains as two launches (first 8 samples; others + frame parts)
    if os.environ.get("SVO_BENCH_HYP_FIRST"):
        ms.set_option("hyp_first", int(os.environ["SVO_BENCH_HYP_FIRST"]))   # RANSAC samples per sequence in a step's first launch (default 8)
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
    probe = list(ms.debug_stream_probe())
    ms.close()
    same = all(m[t, 0].tobytes() == single[t].tobytes() for t in range(msteps))
    # the same with twice the sequences per step (detail only): the rate the step's latency chain no longer bounds
    wide = None
    S2, steps2 = 2 * S, 24
    if (S2 + steps2 + 2) * frame_bytes <= dL.numel():
        ms2 = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=S2)
        ms2.set_option("multi_pipeline", int(os.environ.get("SVO_BENCH_MULTI_PIPELINE", "1")))
        mres2 = torch.zeros((steps2 * S2, rec), dtype=torch.uint8, device=dev)
        for rep in range(2):        # (the first pass allocates and warms up)
            ms2.track_multi_reset(S2, cam)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for t in range(steps2):
                ms2.track_multi_step_dev(dL.data_ptr() + t * frame_bytes, dR.data_ptr() + t * frame_bytes, PITCH, S2, mres2.data_ptr() + t * S2 * rec)
            ms2.sync()
            dt2 = time.perf_counter() - t2
        m2 = mres2.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(steps2, S2)
        ms2.close()
        wide = {"value": steps2 * S2 / dt2, "sequences": S2, "steps": steps2,
                "sequence0_equals_single_chain": bool(all(m2[t, 0].tobytes() == single[t].tobytes() for t in range(steps2)))}
    return {"value": msteps * S / mdt, "value_at_twice_the_sequences": wide, "stream_probe": probe, "unit": "stereo frames/s", "sequences": S, "steps": msteps,
            "sequence0_equals_single_chain": bool(same),
            "pipelined_steps": bool(int(os.environ.get("SVO_BENCH_MULTI_PIPELINE", "1"))),
            "note": "SURVEY 8e's 'G independent sequences' variant on ONE GPU: full Tracking::Track per frame; with "
                    "pipelined_steps the stateless front end of step t + 1 runs beside the tail of step t (multi_pipeline option)"}


def sharded_run(pkg, cam, dL, dR, n_frames, G, devices, rec, chunk=512, reference=None):
    """BASELINE configs[3]: ONE sequence, pair k's front end on context k mod G (context g on devices[g]), the ordered tail
    on context 0 (svo_track_sharded_dev), `chunk` frames per call.  Returns the leg's dict; `reference`: the single-context
    records the result must equal byte for byte."""
    import torch
    per = (chunk + G - 1) // G
    dev0 = torch.device("cuda", devices[0])
    ctxs = [pkg.Svo(W, H, device=devices[g], max_kp=500, max_batch=per) for g in range(G)]
    # pair k resident where its front end runs: context g gets the pairs k = g (mod G) of every chunk, packed in order
    chunks = [(c0, min(chunk, n_frames - c0)) for c0 in range(0, n_frames, chunk)]
    Ls, Rs = [], []
    for g in range(G):
        dg = torch.device("cuda", devices[g])
        idx = torch.cat([torch.arange(c0 + g, c0 + c, G) for c0, c in chunks if c > g])
        Ls.append(dL[idx.to(dL.device)].to(dg)); Rs.append(dR[idx.to(dR.device)].to(dg))
    res = torch.zeros((n_frames, rec), dtype=torch.uint8, device=dev0)
    for d in set(devices):
        torch.cuda.synchronize(d)
    fb = H * PITCH

    def run_all():
        ctxs[0].track_reset(cam)
        off = [0] * G
        for c0, c in chunks:
            pl = [Ls[g].data_ptr() + off[g] * fb for g in range(G)]
            pr = [Rs[g].data_ptr() + off[g] * fb for g in range(G)]
            pkg.Svo.track_sharded_dev(ctxs, pl, pr, PITCH, c, res.data_ptr() + c0 * rec)
            for g in range(G):
                off[g] += (c - g + G - 1) // G
        ctxs[0].sync()
    run_all()                                  # warm-up (allocations, peer mappings)
    t0 = time.perf_counter()
    run_all()
    dt = time.perf_counter() - t0
    got = res.cpu().numpy()
    probe = list(ctxs[0].debug_stream_probe())
    for c in ctxs:
        c.close()
    out = {"value": n_frames / dt, "stream_probe": probe, "unit": "stereo frames/s", "contexts": G, "devices": sorted(set(devices)),
           "frames": int(n_frames), "frames_per_call": chunk,
           "note": "ONE sequence: the stateless front end shards by pair (pair k -> context k mod G), the strict chain of "
                   "src/Tracking.cc:231-250 stays on context 0 and bounds the rate at the tail's per-frame latency whatever G is "
                   "(the front end is ~7 us per pair, the tail ~115 us per frame with the order-preserving EPnP): a FLAT curve over G "
                   "is the most this path can give - replicas (n_gpus independent sequences) are what scales"}
    if reference is not None:
        out["records_identical_to_single_context"] = bool(got.tobytes() == reference.tobytes())
    return out


SEM_ELAS_OPTIONS = {k: int(v) for k, v in (kv.split("=") for kv in filter(None, os.environ.get("SVO_BENCH_SEM_ELAS_OPTIONS", "").split(",")))}   # experiments


def semantic_elas_leg(pkg, cam, dL, dR, dev, rec, n=256):
    """BASELINE configs[4] AS A WHOLE: offline detection boxes (semantic gating: creation gates, brute-force matches ->
    8-point F -> epipolar veto, all on the device) + dense ELAS depth (svo_elas_batch_dev -> disp2Depth -> per-keypoint
    lookups) + the full tracking tail, through svo_track_batch_dev on frames resident in HBM."""
    import numpy as np
    import torch
    n = min(n, dL.shape[0])
    svo = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=n)
    svo.set_option("depth_source", 1)
    for k, v in SEM_ELAS_OPTIONS.items():        # experiments (SVO_BENCH_SEM_ELAS_OPTIONS)
        svo.set_option(k, v)
    bx, keep = boxes_hbm(pkg, n, dev)
    res = torch.zeros((n, rec), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    fps = 0.0
    for _ in range(2):
        svo.track_reset(cam)
        t0 = time.perf_counter()
        svo.track_batch_dev(dL.data_ptr(), dR.data_ptr(), PITCH, n, res.data_ptr(), boxes=bx); svo.sync()
        fps = n / (time.perf_counter() - t0)
    r = res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
    # the same with twice the frames per call when they are resident (the pipeline's fill and drain weigh half as much)
    fps2, n2 = None, 2 * n
    if dL.shape[0] >= n2:
        svo2 = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=n2)
        svo2.set_option("depth_source", 1)
        for k, v in SEM_ELAS_OPTIONS.items():
            svo2.set_option(k, v)
        bx2, keep2 = boxes_hbm(pkg, n2, dev)
        res2 = torch.zeros((n2, rec), dtype=torch.uint8, device=dev)
        for _ in range(2):
            svo2.track_reset(cam)
            t0 = time.perf_counter()
            svo2.track_batch_dev(dL.data_ptr(), dR.data_ptr(), PITCH, n2, res2.data_ptr(), boxes=bx2); svo2.sync()
            fps2 = n2 / (time.perf_counter() - t0)
        same_prefix = bool(res2[:n].cpu().numpy().tobytes() == res.cpu().numpy().tobytes())
        svo2.close()
    # per-kernel times of the same call, one more (untimed) pass with the timers on
    svo.profile_enable(True); svo.profile_reset()
    svo.track_reset(cam)
    svo.track_batch_dev(dL.data_ptr(), dR.data_ptr(), PITCH, n, res.data_ptr(), boxes=bx); svo.sync()
    svo.profile_enable(False)
    prof = svo.profile()
    sem_probe = list(svo.debug_stream_probe())
    svo.close()
    # us per frame: batched kernels by their total over the call, the tail's per-frame kernels (timed on every 32nd frame only)
    # by their average launch
    kern = {k: (v[0] * 1e3 / v[1] if k.startswith(("k_ti_", "k_tp_", "k_tg_")) else v[0] * 1e3 / n)
            for k, v in prof.items() if k.startswith("k_") and v[1] > 0}
    P = W * H
    algo = {"k_elas_desc": 34 * P, "k_elas_match": 48 * P, "k_elas_raster": 8 * P, "k_cc_segments": 28 * P, "k_elas_gap": 16 * P,
            "k_elas_support": 32 * P, "k_elas_lr": 16 * P, "k_elas_mean": 16 * P}
    dom = max((k for k in kern if k in algo), key=lambda k: kern[k], default=None)
    roof = None
    if dom:
        ach = algo[dom] / (kern[dom] * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom, 1), "algorithmic_bytes_per_frame": algo[dom],
                "pipeline_frac": (198.0 * P + ALGO_BYTES_PER_PAIR / 2) * fps / 1e9 / HBM_PEAK_GBS,
                "note": "the dense maps dominate: ELAS's compulsory 198 P bytes per pair + ORB on the left image; the leg is bound by "
                        "ELAS's host stages (support-point filter, Delaunay) under the CPU quota, the gating kernels add "
                        "k_tg_bf + k_tg_fmat per frame (see kernel_us_per_frame)"}
    out = {"value": fps, "stream_probe": sem_probe, "unit": "stereo frames/s", "frames": int(n), "boxes_per_frame": 2,
           "mean_lm_edges": float(r["n_lm_edges"][1:].mean()), "mean_new_mappoints": float(r["n_new_mappoints"].mean()),
           "kernel_us_per_frame": {k: round(v, 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])[:14]},
           "roofline": roof, "host_cpus": host_cpus(),
           "value_at_twice_the_frames_per_call": None if fps2 is None else {"value": fps2, "frames": int(n2), "first_frames_identical_to_the_shorter_call": same_prefix},
           "workload": "BASELINE configs[4]: synth-kitti00 frames with two moving offline boxes per frame (main.cpp:82-95 format), "
                       "depth_source = 1 (dense ELAS map -> disp2Depth -> per-keypoint lookups), full Tracking::Track tail"}
    try:
        from oracle import binding as ob
        ob.build()
        if ob.ref_elas_lib() is not None:
            m = min(64, n)
            Lh = dL[:m, :, :W].cpu().numpy(); Rh = dR[:m, :, :W].cpu().numpy()
            trk = ob.Tracker(W, H, dict(fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, bf=cam.bf))
            t0 = time.perf_counter()
            same = True
            for k in range(m):
                dmap = ob.ref_elas(Lh[k], Rh[k])[0]
                rr, _ = trk.track(Lh[k], Rh[k], boxes=np.array(moving_boxes(k), np.int32), dense=dmap)
                same = same and all(int(rr[f]) == int(r[k][f]) for f in ("n_kp", "n_stereo", "n_match_pass1", "n_match_pass2",
                                                                          "n_lm_edges", "n_new_mappoints", "n_local_map"))
            dt = time.perf_counter() - t0
            trk.close()
            out["cpu_baseline"] = {"value": m / dt, "unit": "stereo frames/s", "cores": 1, "kind": "reference+port",
                                   "sample": "first %d frames: the reference's own compiled libelas (oracle/_ref) for the dense map + the "
                                             "oracle tracker with the same boxes, single thread, %.1f s" % (m, dt),
                                   "counters_identical_to_gpu": bool(same)}
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline_error"] = str(e)
    return out



def host_feed_leg(pkg, cam, dL, dR, dev, local, res_ref, B, calls=12, warm=2):
    """The PCIe-inclusive tracker ("H2D 2 P bytes, D2H ~30 KB" per pair; main.cpp:159-195 hands over one pair per
    Track()): svo_track_batch_host over the first B x calls frames of the headline sequence, the images starting in HOST memory -
    once pinned (copied where they lie), once pageable (staged inside the call) - the records landing in host memory; `warm`
    untimed calls, then the rest back to back with ONE svo_sync at the end.  Beside it the front end alone, host to host."""
    import numpy as np
    import torch
    n = min(B * calls, int(dL.shape[0]))
    calls = n // B
    n = B * calls
    hL = dL[:n].cpu(); hR = dR[:n].cpu()
    fb = H * PITCH
    out = {"frames": int(n), "frames_per_call": int(B), "calls_timed": int(calls - warm), "host_bytes_per_pair": 2 * fb,
           "note": "images start in host memory (pitch %d), uploads on the context's copy stream eight pairs per event, records D2H behind "
                   "the pose chain; no synchronisation between calls" % PITCH}
    t_pin = time.perf_counter()
    pL = hL.pin_memory(); pR = hR.pin_memory()
    out["pin_seconds"] = round(time.perf_counter() - t_pin, 2)
    for name, (aL, aR) in (("pinned", (pL, pR)), ("pageable", (hL, hR))):
        ctx = pkg.Svo(W, H, device=local, max_kp=500, max_batch=B)
        ctx.track_reset(cam)
        res = np.zeros(n, pkg.TRACK_DTYPE)
        for c in range(warm):
            ctx.track_batch_host(aL.data_ptr() + c * B * fb, aR.data_ptr() + c * B * fb, PITCH, B, res[c * B:(c + 1) * B])
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(warm, calls):
            ctx.track_batch_host(aL.data_ptr() + c * B * fb, aR.data_ptr() + c * B * fb, PITCH, B, res[c * B:(c + 1) * B])
        t_enq = time.perf_counter() - t0
        ctx.sync()
        dt = time.perf_counter() - t0
        ctx.close()
        same = bool(res.tobytes() == res_ref[:n].tobytes()) if res_ref is not None else None
        out[name] = {"value": (calls - warm) * B / dt, "unit": "stereo frames/s", "host_ms_in_the_calls": round(1e3 * t_enq, 1),
                     "records_identical_to_the_resident_run": same, "h2d_gb_per_s": (calls - warm) * B * 2 * fb / dt / 1e9}
    # the front end alone, host to host (pinned source): what the link delivers
    Bf = 128
    fe = pkg.Svo(W, H, device=local, max_kp=500, max_batch=Bf, flags=pkg.CREATE_TAIL_ALL_CUS)
    hn = np.zeros(Bf, np.int32); hdepth = np.zeros((Bf, 500), np.float32)
    steps = 16

    def run(s):
        off = (s * Bf) % (n - Bf + 1)
        fe.frontend_batch_host(pL.data_ptr() + off * fb, pR.data_ptr() + off * fb, PITCH, Bf, cam, nL=hn, depth=hdepth)
    for s in range(3):
        run(s)
    fe.sync()
    t0 = time.perf_counter()
    for s in range(3, 3 + steps):
        run(s)
    fe.sync()
    dt = time.perf_counter() - t0
    fe.close()
    out["frontend"] = {"value": Bf * steps / dt, "unit": "stereo pairs/s", "pairs_per_call": Bf, "h2d_gb_per_s": Bf * steps * 2 * fb / dt / 1e9,
                       "note": "svo_frontend_batch_host from pinned memory: bound by the PCIe link, not by the kernels"}
    out["value"] = out["pinned"]["value"]
    out["unit"] = "stereo frames/s"
    del pL, pR
    return out



class Cotenant:
    """Another GPU user of the process beside the tracker (the reference runs its detector on a thread of its own, src/semantic.cc:13-45):
    a host thread that keeps ~1 ms kernels (a float32 matrix product) running back to back on `where` = "null" (the NULL stream, what a
    library that knows nothing of streams uses - torch's default stream) or "pooled" (a non-blocking stream of its own)."""

    def __init__(self, dev, where):
        import torch
        self.torch, self.dev, self.where = torch, dev, where
        self.a = torch.randn((3072, 3072), device=dev)
        self.stream = None if where == "null" else torch.cuda.Stream(device=dev)
        self.stop = threading.Event()
        self.launches = 0
        self.thread = threading.Thread(target=self.run, daemon=True)

    def run(self):
        torch = self.torch
        ev = torch.cuda.Event()
        while not self.stop.is_set():
            if self.stream is None:
                torch.mm(self.a, self.a)
                ev.record()
            else:
                with torch.cuda.stream(self.stream):
                    torch.mm(self.a, self.a)
                    ev.record(self.stream)
            ev.synchronize()                 # one kernel in flight at a time: a loop, not a flood
            self.launches += 1

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.thread.join()


def cotenant_leg(pkg, cam, dL, dR, dev, local, res_ref, B, calls=6, warm=1):
    """The headline tracker (svo_track_batch_dev, B frames per call) while another component of the process keeps the GPU busy - on the
    NULL stream, then on a non-blocking stream of its own - for a context with the default dedicated (blocking) queues and for one
    with pooled non-blocking streams; and two independent trackers (two contexts, two host threads) on the one GPU.  Rates are
    what they are (the co-tenant's kernels fill every CU); the records must not change."""
    import numpy as np
    import torch
    n = min(B * calls, int(dL.shape[0]))
    calls = n // B
    rec = pkg.TRACK_DTYPE.itemsize
    fb = H * PITCH

    def track(flags, out=None, first=0):
        ctx = pkg.Svo(W, H, device=local, max_kp=500, max_batch=B, flags=flags)
        ctx.track_reset(cam)
        res = torch.zeros((n, rec), dtype=torch.uint8, device=dev)
        for c in range(warm):
            ctx.track_batch_dev(dL.data_ptr() + (first + c * B) * fb, dR.data_ptr() + (first + c * B) * fb, PITCH, B, res.data_ptr() + c * B * rec)
        ctx.sync()
        t0 = time.perf_counter()
        for c in range(warm, calls):
            ctx.track_batch_dev(dL.data_ptr() + (first + c * B) * fb, dR.data_ptr() + (first + c * B) * fb, PITCH, B, res.data_ptr() + c * B * rec)
        ctx.sync()
        dt = time.perf_counter() - t0
        ctx.close()
        r = res.cpu().numpy()
        if out is not None:
            out.append(((calls - warm) * B / dt, r))
        return (calls - warm) * B / dt, r

    out = {"frames_per_call": int(B), "calls_timed": int(calls - warm), "unit": "stereo frames/s",
           "cotenant": "a host thread running 3072^3 float32 matrix products (~1 ms each) back to back, one in flight at a time"}
    want = res_ref[:n].tobytes() if res_ref is not None else None
    same = True
    for mode, flags in (("dedicated_queues", 0), ("pooled_streams", pkg.CREATE_POOLED_STREAMS)):
        m = {}
        v, r = track(flags)
        m["alone"] = v
        same = same and (want is None or r.tobytes() == want)
        for where in ("null", "pooled"):
            with Cotenant(dev, where) as ct:
                time.sleep(0.05)
                v, r = track(flags)
            m["beside_%s_stream_cotenant" % where] = v
            m["cotenant_kernels_%s" % where] = ct.launches
            same = same and (want is None or r.tobytes() == want)
        out[mode] = m
    # two independent trackers in one process on one GPU (two contexts, two host threads): each its own sequence
    if int(dL.shape[0]) >= 2 * n:
        res2 = [[], []]
        th = [threading.Thread(target=track, args=(0, res2[k], k * n)) for k in range(2)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        out["two_contexts_one_gpu"] = {"aggregate": sum(r[0][0] for r in res2), "each": [r[0][0] for r in res2],
                                       "first_equals_the_single_tracker": bool(want is None or res2[0][0][1].tobytes() == want)}
        same = same and out["two_contexts_one_gpu"]["first_equals_the_single_tracker"]
    out["records_identical"] = bool(same)
    out["value"] = out["dedicated_queues"]["beside_null_stream_cotenant"]
    return out


PNP_SOLVERS = {
    2: "epnp_exact=2 (default): order-preserving wave EPnP - every IEEE operation of OpenCV's solvePnPRansac/EPnP loops kept, independent "
       "ones spread over a wavefront per RANSAC sample (k-ordered sums on v_mfma_f64_4x4x4); bit-identical per sample to the CPU "
       "restatement, RANSAC outcome identical on all 4,541 frames (tests/test_full_length.py[ord])",
    1: "epnp_exact=1: the same operations one lane per sample, loop by loop (the checker of mode 2)",
    0: "epnp_exact=0: statistical wave EPnP (parallel-order Jacobi, normal equations, FMA): same estimator, another rounding - "
       "RANSAC winner differs on ~3 % of the frames, 4 % of the frames outside 1e-4 m / 1e-5 (tests/test_full_length.py[fast])",
}


def solver_modes_leg(pkg, cam, dL, dR, frame_bytes, rec, dev, n_frames):
    """The headline workload again with the two other EPnP solvers (svo_set_option "epnp_exact"): same frames, same schedule,
    a shorter run each.  Mode 0 was the default up to round 3."""
    import torch
    out = {}
    for mode, frames in ((0, min(n_frames, 2560)), (1, min(n_frames, 512))):
        svo = pkg.Svo(W, H, device=dev.index or 0, max_kp=500, max_batch=256)
        svo.set_option("epnp_exact", mode)
        svo.track_reset(cam)
        res = torch.zeros((frames, rec), dtype=torch.uint8, device=dev)
        B = 128
        def run(c0, c):
            svo.track_batch_dev(dL.data_ptr() + c0 * frame_bytes, dR.data_ptr() + c0 * frame_bytes, PITCH, c, res.data_ptr() + c0 * rec)
        run(0, B); run(B, B)                             # warm-up: two calls (both alternating output sets get allocated)
        svo.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        enq = []
        for c0 in range(2 * B, frames, B):
            run(c0, min(B, frames - c0))
            enq.append(round((time.perf_counter() - t0) * 1e3, 1))
        svo.sync()
        enq.append(round((time.perf_counter() - t0) * 1e3, 1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        enq.append(round(dt * 1e3, 1))
        chain = None
        try:
            chain = tail_chain_from_stamps(svo.debug_track_frames(0, B))
        except Exception:  # noqa: BLE001
            pass
        out.setdefault("stream_probe", {})["epnp_exact=%d" % mode] = list(svo.debug_stream_probe())
        kern = None
        if mode == 0:   # per-kernel times of one more (untimed) call, the timers on
            svo.profile_reset(); svo.profile_enable(True)
            run(0, B); svo.sync()
            svo.profile_enable(False)
            kern = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in svo.profile().items()}
        svo.close()
        out["epnp_exact=%d" % mode] = {"value": (frames - 2 * B) / dt, "unit": "stereo frames/s", "frames_timed": int(frames - 2 * B),
                                        "solver": PNP_SOLVERS[mode],
                                        "host_ms_after_each_call_then_sync": enq, "tail_critical_path": chain, "kernel_avg_us": kern}
    return out


def tail_chain_from_stamps(dbg):
    """The ordered tail's critical path from the in-kernel wall-clock stamps (s_memrealtime, 10 ns ticks) the tail kernels
    leave in every frame's work record: no profiler, no event pairs in the stream."""
    import numpy as np
    rt = dbg["rt"].astype(np.int64)
    # a frame's stamps count only if all five were written by this run, in order, within a second of each other (a kernel that
    # leaves one out - or a stale record - would otherwise turn up as a mean of 1e13 us)
    ok = (rt[:, :5] > 0).all(axis=1) & (rt[:, 3] > rt[:, 4]) & (rt[:, 4] > rt[:, 2]) & (rt[:, 1] > rt[:, 0]) & (rt[:, 3] - rt[:, 2] < 100000000)
    rt = rt[ok]
    if len(rt) < 4:
        return None
    pose_busy = (rt[:, 3] - rt[:, 2]) * 0.01                       # us: first RANSAC workgroup start -> end of the pose kernel
    resolve = (rt[:, 1] - rt[:, 0]) * 0.01
    period = np.diff(rt[:, 3]) * 0.01                              # us between consecutive frames' pose-kernel ends
    period = period[(period > 0) & (period < 5000)]
    hyp = (rt[:, 4] - rt[:, 2]) * 0.01                             # RANSAC samples: first workgroup start -> the frame part's start (two
    frm = (rt[:, 3] - rt[:, 4]) * 0.01                             #   launches: includes the boundary; one launch: the last sample's result seen); frame part start -> end
    rounds = dbg["rounds"][ok]
    return {"frames_sampled": int(len(rt)), "mean_rounds_pass1_pass2": [float(rounds[:, 0].mean()), float(rounds[:, 1].mean())],
            "pose_chain_busy_us": {"mean": float(pose_busy.mean()), "median": float(np.median(pose_busy))},
            "k_ti_resolve_us": {"mean": float(resolve.mean()), "median": float(np.median(resolve)), "max": float(resolve.max())},
            "k_tp_hyp_us": {"mean": float(hyp.mean()), "median": float(np.median(hyp))},
            "k_tp_frame_us": {"mean": float(frm.mean()), "median": float(np.median(frm))},
            "frame_period_us": {"mean": float(period.mean()), "median": float(np.median(period))},
            "source": "in-kernel s_memrealtime stamps of the timed region's last step (svo_debug_track_frames)"}



# ------------------------------------------------------------------------------------------------ the one JSON line
LINE_LIMIT = 4096   # bytes: the driver keeps a bounded tail of the output; a line beyond this was recorded as `parsed: null` (round 5)
LEG_NAMES = ("frontend", "multi_sequence", "sharded", "semantic_elas", "elas", "msa", "host_feed", "host_feed_pageable", "frontend_host_feed",
             "with_null_stream_cotenant", "with_pooled_stream_cotenant", "two_contexts_one_gpu")


def _sig(v, digits=5):
    """floats to `digits` significant digits (what the line carries; the detail file keeps everything)"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        return float("%.*g" % (digits, v))
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, detail_path=None):
    """The line the driver parses, from the full result `out`: the contract's fields, `config` (workload + the accuracy figures),
    `roofline` and `cpu_baseline` as numbers without prose, ONE scalar per leg.  Everything else - notes, per-leg rooflines, kernel
    tables, critical path - is in the detail file (`detail`).  Never longer than LINE_LIMIT bytes: optional parts are dropped in a
    fixed order until it fits (tests/test_bench_line.py runs this on the canned full results under profiles/)."""
    cfg = out.get("config", {}) or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    c = _pick(cfg, ("workload", "parallelism", "sharding", "contexts", "devices", "pairs_per_step_per_gpu", "resident_pairs_per_gpu",
                    "resident_input_bytes_per_gpu", "concurrent_sequences_per_gpu", "frames_tracked", "ate_rmse_m_vs_ground_truth",
                    "final_position_error_m", "path_length_m", "mean_lm_edges", "tracker_capacity_flag",
                    "pnp_samples_through_sequential_fallback", "mean_keypoints_left", "mean_stereo_depths"))
    if isinstance(c.get("workload"), str) and len(c["workload"]) > 330:
        c["workload"] = c["workload"][:327] + "..."
    ate = out.get("ate_vs_cpu_m")
    if isinstance(ate, dict):
        c["ate_vs_cpu_m_rmse"] = ate.get("rmse")
        c["ate_vs_cpu_m_max"] = ate.get("max")
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("counters_identical_to_gpu") is not None:
        c["counters_identical_to_cpu"] = cb.get("counters_identical_to_gpu")
    line["config"] = c
    rf = out.get("roofline")
    if isinstance(rf, dict):
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                       "kernel_seconds_per_launch", "pipeline_frac", "share_of_kernel_time_per_frame"))
        r.setdefault("traffic", None)
        im = rf.get("issue_model")
        if isinstance(im, dict):   # the model that governs the critical-path kernel: numbers only
            r["issue_model"] = _pick(im, ("peak_inst_per_quad_cycle_per_wave", "achieved_inst_per_quad_cycle_per_wave",
                                          "instructions_per_wave", "share_of_frame_period"))
        line["roofline"] = r
    if isinstance(cb, dict):
        b = _pick(cb, ("value", "unit", "cores", "kind", "sample"))
        if isinstance(b.get("sample"), str) and len(b["sample"]) > 200:
            b["sample"] = b["sample"][:197] + "..."
        line["cpu_baseline"] = b
    for name in LEG_NAMES:
        leg = out.get(name)
        if isinstance(leg, dict):
            line[name] = leg.get("value", leg.get("error"))
        elif leg is not None:
            line[name] = leg
    # the legs' own parity checks (booleans; the detail file says what each compared)
    checks = {}
    for name, key, path in (("sharded_records_identical_to_single_context", "sharded", ("records_identical_to_single_context",)),
                            ("multi_sequence0_equals_single_chain", "multi_sequence", ("sequence0_equals_single_chain",)),
                            ("host_feed_records_identical_to_resident", "host_feed", ("pinned", "records_identical_to_the_resident_run")),
                            ("host_feed_pageable_records_identical", "host_feed", ("pageable", "records_identical_to_the_resident_run")),
                            ("semantic_elas_counters_identical_to_cpu", "semantic_elas", ("cpu_baseline", "counters_identical_to_gpu")),
                            ("elas_pixels_differing_from_reference", "elas", ("pixels_differing_from_reference",)),
                            ("msa_pixels_differing_from_port", "msa", ("pixels_differing_from_port",)),
                            ("cotenant_records_identical", "with_null_stream_cotenant", ("records_identical",))):
        v = out.get(key)
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        if v is not None:
            checks[name] = v
    if checks:
        line["checks"] = checks
    tc = out.get("tail_critical_path")
    if isinstance(tc, dict) and "frame_period_us" in tc:
        line["frame_period_us"] = tc["frame_period_us"].get("mean")
    if detail_path:
        line["detail"] = detail_path
    exact = {k: line.get(k) for k in ("value", "ms_per_step")}
    line = _sig(line)
    for k, v in exact.items():      # the two figures the driver cross-checks keep nine digits
        if isinstance(v, float):
            line[k] = float("%.9g" % v)
    # shrink in a fixed order if a future field pushed it over the limit
    for drop in (("frame_period_us",), ("roofline", "issue_model"), ("cpu_baseline", "sample"), ("config", "parallelism"),
                 ("config", "resident_input_bytes_per_gpu"), ("config", "final_position_error_m"), ("config", "path_length_m")):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k, {}) if isinstance(tgt, dict) else {}
        if isinstance(tgt, dict):
            tgt.pop(drop[-1], None)
    if len(json.dumps(line)) >= LINE_LIMIT and isinstance(line.get("config", {}).get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:120]
    return line


DETAIL_PATH = None     # --detail-path


def emit(out):
    """Full result -> bench_detail.json beside this file (fallback: the temp directory); the compact line -> stdout, LAST."""
    path = DETAIL_PATH or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        import tempfile
        path = os.path.join(tempfile.gettempdir(), "bench_detail.json")
        try:
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            path = None
    sys.stderr.flush()
    text = json.dumps(compact_line(out, os.path.basename(path) if path else None))
    assert len(text) < LINE_LIMIT, "bench line of %d bytes" % len(text)
    print(text)
    sys.stdout.flush()


# ------------------------------------------------------------------------------------------------ main
def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a CHILD process group before this process makes
    any GPU call (it never initialises HIP), relay rank 0's JSON line, exit with the child's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            print(line)
    sys.stdout.flush()
    sys.exit(p.returncode)


TAIL_LEGS = {"pnp_solver_modes": 1280, "sharded": 1024, "multi_sequence": max(128, int(os.environ.get("SVO_BENCH_MULTI_S", "64")) + 64), "semantic_elas": 512}   # leg -> frames it renders


def tail_leg_reference(pkg, cam, dL, dR, dev, local, nr):
    """the single context's records of the first `nr` frames (what `sharded` and `multi_sequence` must reproduce)"""
    import torch
    rec = pkg.TRACK_DTYPE.itemsize
    fb = H * PITCH
    ref = torch.zeros((nr, rec), dtype=torch.uint8, device=dev)
    s = pkg.Svo(W, H, device=local, max_kp=500, max_batch=256)
    s.track_reset(cam)
    for c0 in range(0, nr, 256):
        s.track_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, PITCH, min(256, nr - c0), ref.data_ptr() + c0 * rec)
    s.sync(); s.close()
    return ref.cpu().numpy()


def run_tail_leg(name, pkg, cam, dL, dR, dev, local, refn=None):
    """One of the legs whose figure depends on the tracker's index chain and pose chain running side by side (solver modes,
    sharded, many sequences, configs[4]) on the frames resident in dL / dR (>= TAIL_LEGS[name] of them)."""
    rec = pkg.TRACK_DTYPE.itemsize
    fb = H * PITCH
    N = TAIL_LEGS[name]
    if refn is None and name in ("sharded", "multi_sequence"):
        refn = tail_leg_reference(pkg, cam, dL, dR, dev, local, min(1024, int(dL.shape[0])))
    try:
        if name == "pnp_solver_modes":
            r = solver_modes_leg(pkg, cam, dL, dR, fb, rec, dev, N)
        elif name == "sharded":
            r = sharded_run(pkg, cam, dL, dR, 1024, 2, [local, local], rec, reference=refn[:1024])
            r["note"] = ("two contexts on this ONE GPU: no cross-device run was measured here (the driver's multi-GPU run, "
                         "when it has a node, adds it); " + r["note"])
        elif name == "multi_sequence":
            r = multi_sequence_leg(pkg, cam, dL, dR, fb, rec, dev, refn[:48].view(pkg.TRACK_DTYPE).reshape(-1))
        else:
            r = semantic_elas_leg(pkg, cam, dL, dR, dev, rec)
    except Exception as e:  # noqa: BLE001
        r = {"error": repr(e)}
    return r


def leg_value(r):
    """a leg's figure: its "value", or {sub-leg: value} (pnp_solver_modes), or its error"""
    if not isinstance(r, dict):
        return r
    if "value" in r:
        return r["value"]
    sub = {k: v["value"] for k, v in r.items() if isinstance(v, dict) and "value" in v}
    return sub or r.get("error")


def leg_ratio(a, b):
    try:
        if isinstance(a, dict):
            return {k: a[k] / b[k] for k in a}
        return a / b
    except Exception:  # noqa: BLE001
        return None


def tail_leg_child(args, pkg, synth, cam, dev, local):
    """One tail leg in a process of its own (`--tail-leg-child <leg>`): the figure a process that does nothing else gets.  Up
    to round 4 this was the only way these legs measured their full rate (two streams of a context could end up on one
    hardware queue, depending on the process's earlier streams); bench.py now runs them in-process, in two orders, and
    prints this figure beside them.  Prints one JSON object {leg: result}."""
    name = args.tail_leg_child
    dL, dR, _ = render_frames(synth, TAIL_LEGS[name], dev, synth.BASE_SEED)
    progress("leg %s (child process)" % name)
    print(json.dumps({name: run_tail_leg(name, pkg, cam, dL, dR, dev, local)}))
    sys.stdout.flush()


def progress(msg):
    """One line per stage on stderr (the JSON line is stdout's): a run that stops answering shows where."""
    sys.stderr.write("[bench %.1f s] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


def start_watchdog(seconds):
    """A run that stops answering must not sit until the driver's limit: after `seconds` the process says where it was and
    exits non-zero (no re-exec: the GPU may be in use).  One unexplained stall was seen in round 4 (never again in >40 runs)."""
    import threading

    def fire():
        sys.stderr.write("[bench %.1f s] watchdog: no result after %d s - giving up (exit code 3)\n" % (time.perf_counter() - T_START, seconds))
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="track", choices=["track", "frontend"])
    ap.add_argument("--frames", type=int, default=4541, help="length of the resident sequence (KITTI 00: 4541)")
    ap.add_argument("--batch", type=int, default=0,
                    help="pairs per step per GPU: track default frames // (warmup + steps), frontend default 128")
    ap.add_argument("--watchdog", type=int, default=1500, help="seconds after which a run that has not printed its line exits with code 3 (0: off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the untimed per-kernel profile pass")
    ap.add_argument("--no-legs", action="store_true", help="only the main workload (no frontend / multi_sequence / sharded / semantic_elas / elas / msa legs)")
    ap.add_argument("--no-elas-leg", action="store_true", help="skip the semantic_elas, elas and msa legs")
    ap.add_argument("--no-track-leg", action="store_true", help="(kept for scripts) same as --no-legs for the frontend workload")
    ap.add_argument("--tail-leg-child", default=None, choices=sorted(TAIL_LEGS),
                    help="run ONE of the legs that depend on the tracker's two overlapping chains in this (fresh) process and print it")
    ap.add_argument("--leg-orders", action="store_true", help="repeat the tail legs in the reverse order and each in a child process (three figures per leg)")
    ap.add_argument("--quick-legs", action="store_true", help="skip the pnp_solver_modes leg")
    ap.add_argument("--no-tail-leg-children", action="store_true", help="(kept for scripts; the child repetition now needs --leg-orders)")
    ap.add_argument("--no-host-feed-leg", action="store_true", help="skip the host_feed leg (svo_track_batch_host / svo_frontend_batch_host from host memory)")
    ap.add_argument("--no-cotenant-leg", action="store_true", help="skip the co-tenant leg (tracker beside another GPU user of the process; two contexts on one GPU)")
    ap.add_argument("--detail-path", default=None, help="where the full result goes (default: bench_detail.json beside bench.py)")
    ap.add_argument("--no-shard-leg", action="store_true", help="N > 1: skip rank 0's svo_track_sharded_dev run across the N GPUs")
    ap.add_argument("--shard", action="store_true",
                    help="track workload: ONE sequence over --gpus G contexts in ONE process (svo_track_sharded_dev, BASELINE configs[3])")
    ap.add_argument("--depth-source", type=int, default=0, choices=[0, 1, 2],
                    help="track workload: 0 sparse epipolar stereo (north star), 1 dense ELAS map, 2 dense MSA map")
    ap.add_argument("--boxes", action="store_true", help="track workload: two moving offline detection boxes per frame (semantic gating)")
    ap.add_argument("--sequences", type=int, default=1,
                    help="track workload: S concurrent sequences per GPU (svo_track_multi_step_dev), one frame of each per step")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default) or gloo (single-GPU dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks / contexts use cuda:0 (dry run of the N>1 path on one GPU)")
    args = ap.parse_args()
    global DETAIL_PATH
    DETAIL_PATH = args.detail_path
    if args.watchdog > 0:
        start_watchdog(args.watchdog)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and not args.shard and env_world is None:
        spawn_ranks(args)                              # does not return
    if env_world is not None and int(env_world) != args.gpus and not args.shard:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s ranks (WORLD_SIZE): refusing to print a line that "
                         "misstates n_gpus" % (args.gpus, env_world))

    import numpy as np
    import torch
    import svo_loader
    pkg = svo_loader.load()
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    shard = importlib.import_module("stereo_semantic_vo_amd.shard")
    rank, world, local = shard.env_rank_world()
    if args.shard:
        rank, world, local = 0, 1, 0
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
    if args.tail_leg_child:
        tail_leg_child(args, pkg, synth, cam, dev, local)
        return
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

    # ---------------------------------------------------------------- --shard: configs[3] in one process
    if args.shard:
        G = args.gpus
        ndev = torch.cuda.device_count()
        devices = [g if (g < ndev and not args.share_gpu) else 0 for g in range(G)]
        single = pkg.Svo(W, H, device=0, max_kp=500, max_batch=512)
        single.track_reset(cam)
        ref = torch.zeros((n_frames, rec), dtype=torch.uint8, device=dev)
        for c0 in range(0, n_frames, 512):
            c = min(512, n_frames - c0)
            single.track_batch_dev(dL.data_ptr() + c0 * frame_bytes, dR.data_ptr() + c0 * frame_bytes, PITCH, c, ref.data_ptr() + c0 * rec)
        single.sync(); single.close()
        leg = sharded_run(pkg, cam, dL, dR, n_frames, G, devices, rec, reference=ref.cpu().numpy())
        res = ref.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
        rmse, last = ate_rmse(res, T_gt.numpy())
        out = {"metric": "stereo frames/sec on KITTI 00 (full Tracking::Track chain)", "value": leg["value"], "unit": "stereo frames/s",
               "n_gpus": len(set(devices)), "steps": 1, "warmup": 1, "ms_per_step": 1e3 * n_frames / leg["value"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "BASELINE configs[3]: ONE synth-kitti00 sequence of %d pairs, pair k's front end on context k mod %d, "
                                      "ordered tail on context 0 (svo_track_sharded_dev), no collective" % (n_frames, G),
                          "contexts": G, "devices": devices, "frames_tracked": int(n_frames), "ate_rmse_m_vs_ground_truth": rmse,
                          "parallelism": "front end sharded by pair over %d contexts, tail on one" % G},
               "sharded": leg}
        emit(out)
        return

    svo = pkg.Svo(W, H, device=local, max_kp=500, max_batch=B + extra)
    epnp_mode = 2                                        # the library's default: the bit-comparable solver
    if "SVO_BENCH_EPNP" in os.environ:                   # experiments: the headline with another EPnP solver (named in config)
        epnp_mode = int(os.environ["SVO_BENCH_EPNP"])
        svo.set_option("epnp_exact", epnp_mode)
    for kv in filter(None, os.environ.get("SVO_BENCH_OPTIONS", "").split(",")):   # experiments: "track_group=8,..."
        k, v = kv.split("=")
        svo.set_option(k, int(v))
        if k == "epnp_exact":
            epnp_mode = int(v)
    epnp_mode = 0 if epnp_mode == 0 else 2 if epnp_mode == 2 else 1   # (the library's own mapping of the option's values)
    bx = keep_bx = None
    if multi:
        d_res = torch.zeros((nsteps * B, rec), dtype=torch.uint8, device=dev)
        svo.track_multi_reset(B, cam)
    elif track:
        if args.depth_source:
            svo.set_option("depth_source", args.depth_source)
        if args.boxes:
            bx, keep_bx = boxes_hbm(pkg, n_frames, dev)
        d_res = torch.zeros((n_frames, rec), dtype=torch.uint8, device=dev)
        svo.track_reset(cam)
    else:
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        d_depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step_boxes(off):
        if bx is None:
            return None
        return pkg.boxes_dev(keep_bx[0].data_ptr() + off * 2 * 16, keep_bx[1].data_ptr() + off * 4, 2)

    def step(s, ctx=None, out=None):
        ctx = ctx or svo
        if multi:
            ctx.track_multi_step_dev(dL.data_ptr() + s * frame_bytes, dR.data_ptr() + s * frame_bytes, PITCH, B,
                                     d_res.data_ptr() + s * B * rec)
        elif track:
            off = s * B + (extra if s > 0 else 0)
            nb = B + (extra if s == 0 else 0)
            ctx.track_batch_dev(dL.data_ptr() + off * frame_bytes, dR.data_ptr() + off * frame_bytes,
                                PITCH, nb, (out if out is not None else d_res).data_ptr() + off * rec, boxes=step_boxes(off))
        else:
            off = (s * B) % (n_frames - B + 1)           # every step reads B pairs it has not touched for a long time
            ctx.frontend_batch_dev(dL.data_ptr() + off * frame_bytes, dR.data_ptr() + off * frame_bytes, PITCH, B, cam,
                                   d_nL=d_n.data_ptr(), d_depth=d_depth.data_ptr())

    def fence():
        svo.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    progress("warm-up")
    for s in range(args.warmup):
        step(s)
    fence()
    progress("timed region")
    # the timed region: the product's default schedule, no profiling hooks
    t0 = time.perf_counter()
    for s in range(args.warmup, nsteps):
        step(s)
    fence()
    dt = time.perf_counter() - t0
    progress("timed region done")
    dt = shard.max_over_ranks(dt, dist, dev if args.dist_backend == "nccl" else "cpu")
    # in-kernel wall-clock stamps of the last timed step's frames (always written, free)
    chain = None
    if track and not multi and rank == 0:
        try:
            chain = tail_chain_from_stamps(svo.debug_track_frames(0, B))
        except Exception as e:  # noqa: BLE001
            chain = {"error": repr(e)}
    # per-kernel HIP-event times: a separate, UNTIMED pass with the timers on (they change the schedule: one stream for the
    # front end, an event pair around every 32nd frame's tail kernels) over the first frames of the sequence again
    prof = {}
    if not args.no_profile and rank == 0:
        psvo = pkg.Svo(W, H, device=local, max_kp=500, max_batch=B + extra)
        scratch = torch.zeros_like(d_res) if track else None
        if multi:
            psvo.track_multi_reset(B, cam)
        elif track:
            if args.depth_source:
                psvo.set_option("depth_source", args.depth_source)
            psvo.track_reset(cam)
        psvo.profile_reset(); psvo.profile_enable(True)
        for s in range(min(nsteps, 3)):
            step(s, ctx=psvo, out=scratch)
        psvo.sync()
        psvo.profile_enable(False)
        prof = psvo.profile()
        psvo.close()
        del scratch

    if rank == 0:
        pairs = world * B * args.steps
        cfg = {"pairs_per_step_per_gpu": B, "resident_pairs_per_gpu": int(n_frames),
               "resident_input_bytes_per_gpu": int(2 * n_frames * frame_bytes), "render_seconds": round(render_s, 1)}
        res = None
        oracle_run = None
        if track:
            res = d_res.cpu().numpy().view(pkg.TRACK_DTYPE).reshape(-1)
            if multi:                                    # accuracy figures from sequence 0 (starts at frame 0)
                res = res.reshape(-1, B)[:, 0]
                cfg["concurrent_sequences_per_gpu"] = B
            rmse, last = ate_rmse(res, T_gt.numpy()[:len(res)])
            depth_names = {0: "sparse epipolar stereo", 1: "dense ELAS map (svo_elas_batch_dev) -> disp2Depth -> per-keypoint lookups",
                           2: "dense MSA map (svo_msa_batch_dev) -> disp2Depth -> per-keypoint lookups"}
            cfg.update({"depth_source": depth_names[args.depth_source], "detection_boxes": "two moving boxes per frame" if args.boxes else "none",
                        "workload": "BASELINE configs[2]: synth-kitti00 sequence of %d distinct 1241x376 stereo pairs, full "
                                    "Tracking::Track loop per frame (ORB on L and R, sparse stereo, matching passes 1+2, "
                                    "PnP-RANSAC, pose-only LM, map-point lifecycle); " % len(res) +
                                    ("%d concurrent staggered sequences per GPU, one frame of each per step" % B if multi
                                     else "ONE sequence per GPU in strict frame order (replicas across GPUs)"),
                        "parallelism": "replicas: %d independent sequences, one per GPU" % world if world > 1 else "one GPU",
                        "frames_tracked": int(len(res)), "ate_rmse_m_vs_ground_truth": rmse,
                        "final_position_error_m": last, "path_length_m": float(len(res) - 1),
                        "mean_lm_edges": float(res["n_lm_edges"][1:].mean()),
                        "mean_active_rows_pass1_pass2": [float(res["reserved"][1:, 0].mean()), float(res["reserved"][1:, 1].mean())],
                        "mean_local_map": float(res["n_local_map"][1:].mean()),
                        "tracker_capacity_flag": int(svo.track_overflowed()),
                        "pnp_solver": PNP_SOLVERS[epnp_mode],
                        "pnp_samples_through_sequential_fallback": int(svo.track_epnp_fallbacks())})
            if world == 1 and not multi and not args.no_cpu_baseline and args.depth_source == 0 and not args.boxes:
                # the oracle's tail over ALL tracked frames, on a host thread while the legs below run
                oracle_run = OracleTailRun(cam, *frontend_outputs(pkg, cam, dL, dR, len(res), dev))
                oracle_run.start()
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
        if track and not multi:
            try:
                cfg["stream_probe"] = {"candidates_tried": int(svo.debug_stream_probe()[0]), "two_chains_vs_one_percent": int(svo.debug_stream_probe()[1])}
            except Exception:  # noqa: BLE001
                pass
        if chain:
            out["tail_critical_path"] = chain
        if prof:
            kern = {k: {"avg_ms": v[0] / max(v[1], 1), "launches": v[1], "total_ms": v[0]} for k, v in prof.items()}
            # time per frame of the timed region each kernel accounts for: the tail kernels run once per frame (timed on every
            # 32nd frame of the profile pass only: an event pair holds the chain up), the front-end kernels once per sub-batch
            frames_prof = float(sum((B + (extra if s == 0 else 0)) for s in range(min(nsteps, 3)))) if track and not multi else float(B * min(nsteps, 3))
            per_frame = {k: (kern[k]["avg_ms"] if track and not multi and k.startswith(("k_ti_", "k_tp_", "k_tg_")) else kern[k]["total_ms"] / frames_prof)
                         for k in kern}
            for k in kern:
                kern[k]["ms_per_frame"] = per_frame[k]
            # the tail's kernels: in-kernel wall-clock stamps of the timed region where they exist (an event pair around a ~40 us
            # kernel of single-wave workgroups inflates it by up to 10 us; the stamps agree with rocprofv3's averages)
            stamped = {}
            fused_tail = "k_tp_tail_ord" in per_frame      # samples + frame part in one launch (the default)
            hyp_name = "k_tp_tail_ord" if fused_tail else next((k for k in ("k_tp_hyp_ord", "k_tp_hyp", "k_tp_hyp_exact") if k in per_frame), "k_tp_hyp_ord")
            if chain and "k_ti_resolve_us" in chain:
                stamped = {name: chain[key + "_us"]["mean"] * 1e-3 for key, name in
                           (("k_ti_resolve", "k_ti_resolve"), ("k_tp_hyp", hyp_name), ("k_tp_frame", "k_tp_frame")) if name in per_frame}
                if fused_tail:
                    stamped["k_tp_tail_ord"] = (chain["k_tp_hyp_us"]["mean"] + chain["k_tp_frame_us"]["mean"]) * 1e-3
                for k, v in stamped.items():
                    per_frame[k] = v
                    kern[k]["ms_per_frame"] = v
                    kern[k]["ms_per_frame_source"] = "in-kernel stamps"
            dom = max(per_frame.items(), key=lambda kv: kv[1])[0]
            dom_s = kern[dom]["avg_ms"] * 1e-3
            timing = "HIP events around the kernel's launches in an untimed profile pass"
            if dom in stamped:
                dom_s = stamped[dom] * 1e-3
                timing = "in-kernel wall-clock stamps over the timed region's last step (event pairs inflate a ~40 us kernel)"
            units = "images"
            ppl = min(B, 32) if track else B     # pairs one front-end launch covers (svo_track_batch_dev: sub-batches of 32)
            if dom in KERNEL_BYTES_PER_IMAGE:
                algo = KERNEL_BYTES_PER_IMAGE[dom] * 2 * ppl
            elif dom in KERNEL_BYTES_PER_PAIR:
                algo = KERNEL_BYTES_PER_PAIR[dom] * ppl
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
                               "kernel_seconds_per_launch": dom_s, "timing": timing,
                               "share_of_kernel_time_per_frame": per_frame[dom] / max(sum(per_frame.values()), 1e-12),
                               "valu": pmc_valu(dom),
                               "pipeline_frac": ALGO_BYTES_PER_PAIR * (pairs / dt / world) / 1e9 / HBM_PEAK_GBS,
                               "note": "the ordered tail is two dependent chains of small kernels (index chain: k_ti_lists -> k_ti_resolve; "
                                       "pose chain: RANSAC samples -> LM): latency-bound, far from the HBM roof by construction "
                                       "(DESIGN.md section 5); `tail_critical_path` has the chains' in-kernel times" if units == "frames" else None}
            out["kernels"] = kern
            if track and not multi and chain and "k_tp_hyp_us" in chain:
                # the kernel the frame rate hangs on is not the one furthest below the HBM roof: the pose chain (RANSAC samples ->
                # LM) is a dependent float64 instruction stream, one wavefront per sample, and a wavefront issues at most one
                # instruction per 4 cycles whatever the instruction is
                hyp_us, frm_us = chain["k_tp_hyp_us"]["mean"], chain["k_tp_frame_us"]["mean"]
                v = pmc_valu(hyp_name)
                if v and fused_tail:
                    # a fused launch is 101 workgroups of four waves; 300 of its 404 waves leave at once (only the first wave of a
                    # sample workgroup solves): per-wave figures over the 104 waves that work
                    d = pmc_entry(hyp_name)
                    v["instr_all_per_wave"] = (d["SQ_INSTS_VALU"] + d.get("SQ_INSTS_SALU", 0.0) + d.get("SQ_INSTS_LDS", 0.0)) / 104.0
                    v["waves_that_work"] = 104
                out["critical_path"] = {
                    "kernels": ["k_tp_tail_ord: RANSAC samples", "k_tp_tail_ord: frame part (RANSAC rule, LM, new map points)"] if fused_tail
                               else [hyp_name, "k_tp_frame"], "us_per_frame": [hyp_us, frm_us],
                    "share_of_frame_period": (hyp_us + frm_us) / max(chain["frame_period_us"]["mean"], 1e-9),
                    "bound": "float64 instruction issue of ONE wavefront per RANSAC sample (quad-cycle cadence: <= 0.25 instructions per "
                             "cycle per wave): the solve is a dependent chain - 148 Jacobi steps of ~150 instructions (division and "
                             "square root sequences of the rotation: 46 of them) + 5 Gauss-Newton steps per candidate",
                    "issue_model": {"peak_inst_per_quad_cycle_per_wave": 1.0,
                                    "achieved_inst_per_quad_cycle_per_wave": (v or {}).get("inst_per_quad_cycle_per_wave"),
                                    "instructions_per_wave": (v or {}).get("instr_all_per_wave"),
                                    "note": "the rest of the issue slots are dependency stalls of the float64 chain (a dependent v_fma_f64 "
                                            "issues every second quad-cycle, v_rcp / v_rsq_f64 every fourth, a 4x4x4 DMFMA occupies five), "
                                            "LDS round trips of the row exchange and hazard wait states",
                                    "source": "profiles/pmc_latest.json: (SQ_INSTS_VALU + SQ_INSTS_SALU + SQ_INSTS_LDS) / SQ_WAVE_CYCLES of the "
                                              "kernel, rocprofv3 --pmc passes of profiles/r06_z_track_*" if v else "no counters committed for this kernel"},
                    "hbm_frac_of_this_kernel": (tail_kernel_bytes("k_tp_hyp_ord", 0, 0, float(res["n_lm_edges"][1:].mean()), 0) / (hyp_us * 1e-6) / 1e9 / HBM_PEAK_GBS),
                }
                # the headline's roofline leads with the model that governs the critical-path kernel; the HBM fraction (the contract's
                # fields above) comes second
                out["roofline"]["issue_model"] = dict(out["critical_path"]["issue_model"], bound=out["critical_path"]["bound"],
                                                      share_of_frame_period=out["critical_path"]["share_of_frame_period"])
        if world == 1 and not args.no_cpu_baseline:
            ns = min(n_frames, 64)
            Lh = dL[:ns, :, :W].cpu().numpy()
            Rh = dR[:ns, :, :W].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(Lh, Rh, cam, args.workload)
        legs = world == 1 and not args.no_legs and not multi and args.depth_source == 0 and not args.boxes
        if legs:
            svo.close()                          # (its buffers: the legs' contexts bring their own)
            svo = None
        # The legs that run the ordered tail (two chains that must overlap on two hardware queues) run in THIS process, after the
        # headline's context and after each other.  --leg-orders repeats them in the reverse order and each in a process of its own
        # (round 5's finding - all three agree since the context's streams are four hardware queues made back to back - does not
        # need re-proving in every headline run; profiles/r05_leg_orders.jsonl).
        if legs and track:
            names = [n for n in TAIL_LEGS if not (n == "semantic_elas" and args.no_elas_leg) and not (n == "pnp_solver_modes" and not args.leg_orders and args.quick_legs)]
            refn = tail_leg_reference(pkg, cam, dL, dR, dev, local, 1024)
            first, second = {}, {}
            for name in names:
                progress("leg %s: in-process" % name)
                first[name] = run_tail_leg(name, pkg, cam, dL, dR, dev, local, refn)
            if args.leg_orders:
                for name in reversed(names):
                    progress("leg %s: in-process, reverse order" % name)
                    second[name] = run_tail_leg(name, pkg, cam, dL, dR, dev, local, refn)
            for name in names:
                r = first[name]
                if args.leg_orders:
                    r["order"] = "in-process, %d. of %s" % (names.index(name) + 1, " > ".join(names))
                    r["value_in_process"] = leg_value(first[name])
                    r["value_in_process_reverse_order"] = leg_value(second[name])
                    r["in_process_reverse_order"] = {k: v for k, v in second[name].items() if k in ("stream_probe", "epnp_exact=0", "error", "chain_us_per_frame")}
                    progress("leg %s: child process" % name)
                    try:
                        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--tail-leg-child", name],
                                            stdout=subprocess.PIPE, timeout=600, check=True)
                        child = json.loads(cp.stdout.decode().strip().splitlines()[-1])[name]
                        r["value_child_process"] = leg_value(child)
                    except Exception as e:  # noqa: BLE001
                        r["value_child_process"] = repr(e)
                    r["in_process_over_child"] = [leg_ratio(r["value_in_process"], r["value_child_process"]),
                                                  leg_ratio(r["value_in_process_reverse_order"], r["value_child_process"])]
                out[name] = r
            if not args.no_host_feed_leg:
                progress("leg host_feed")
                try:
                    hf = host_feed_leg(pkg, cam, dL, dR, dev, local, res, B)
                    out["host_feed"] = hf
                    out["host_feed_pageable"] = hf["pageable"]["value"]
                    out["frontend_host_feed"] = hf["frontend"]["value"]
                except Exception as e:  # noqa: BLE001
                    out["host_feed"] = {"error": repr(e)}
            if not args.no_cotenant_leg:
                progress("leg cotenant")
                try:
                    ct = cotenant_leg(pkg, cam, dL, dR, dev, local, res, B)
                    out["with_null_stream_cotenant"] = ct
                    out["with_pooled_stream_cotenant"] = ct["dedicated_queues"]["beside_pooled_stream_cotenant"]
                    if "two_contexts_one_gpu" in ct:
                        out["two_contexts_one_gpu"] = ct["two_contexts_one_gpu"]["aggregate"]
                except Exception as e:  # noqa: BLE001
                    out["with_null_stream_cotenant"] = {"error": repr(e)}
            progress("leg frontend")
            out["frontend"] = frontend_leg(pkg, cam, dL, dR, n_frames, frame_bytes, dev,
                                           None if args.no_cpu_baseline else cpu_baseline_all_cores)
        if oracle_run is not None:
            oracle_run.join()
            rep = oracle_run.report(res)
            out["cpu_baseline"]["all_frames_tail"] = rep
            out["cpu_baseline"]["counters_identical_to_gpu"] = rep.get("counters_identical_to_gpu")
            try:   # where the CPU port's time goes: the ordered tail alone (all frames) against the full loop (the 64-frame sample)
                tail_s = float(rep["seconds"]) / max(int(rep["frames"]), 1)
                full_s = 1.0 / out["cpu_baseline"]["value"]
                out["cpu_baseline"]["cpu_seconds_per_frame"] = {"full_loop": full_s, "ordered_tail": tail_s,
                                                                "front_end_share": max(0.0, 1.0 - tail_s / full_s)}
            except Exception:  # noqa: BLE001
                pass
            if "ate_vs_cpu_m" in rep:
                out["ate_vs_cpu_m"] = rep["ate_vs_cpu_m"]
        if legs and not args.no_elas_leg:
            progress("leg elas")
            out["elas"] = elas_leg(pkg, dev.index or 0, dL, dR, PITCH, min(512, int(n_frames)), iters=2)
            progress("leg msa")
            out["msa"] = msa_leg(pkg, dev.index or 0, dL, dR)
    # N > 1: after the replicas' timed region rank 0 alone drives ONE sequence over all N GPUs (BASELINE configs[3]);
    # the other ranks wait at the barrier below
    if world > 1 and rank == 0 and track and not multi and not args.no_shard_leg:
        # In a PROCESS OF ITS OWN, with a time limit: the cross-device paths of svo_track_sharded_dev (peer copies, events of one
        # device waited for on another, per-device contexts) have only ever run with every context on ONE GPU - the builder's boxes
        # have one; whatever happens there on a real node must not take the replicas' line with it.
        try:
            import tempfile
            ns = min(n_frames, 2048)
            tmp = os.path.join(tempfile.gettempdir(), "bench_shard_detail_%d.json" % os.getpid())
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                       "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            cmd = [sys.executable, os.path.abspath(__file__), "--shard", "--gpus", str(world), "--frames", str(ns), "--detail-path", tmp,
                   "--watchdog", "540"] + (["--share-gpu"] if args.share_gpu else [])
            progress("leg sharded: one sequence over %d contexts, child process" % world)
            cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if cp.returncode != 0:
                out["sharded"] = {"error": "child exited with %d: %s" % (cp.returncode, cp.stderr.decode(errors="replace")[-400:])}
            else:
                out["sharded"] = json.load(open(tmp))["sharded"]
            try:
                os.remove(tmp)
            except OSError:
                pass
        except Exception as e:  # noqa: BLE001
            out["sharded"] = {"error": repr(e)}
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if svo is not None:
        svo.close()


if __name__ == "__main__":
    main()
