#!/usr/bin/env python3
"""bench.py - stereo pairs/s through the MI355X-native tracking front end.

One "step" = one pass of the hot path over one batch of B synthetic KITTI-00-shaped
stereo pairs (1241x376 gray) that are already resident in HBM.  Workloads:
  frontend : ORB pyramid extraction on both images + sparse epipolar stereo
             (BASELINE.json configs[1], batched)
  track    : frontend + the ordered tracking tail (matching, PnP-RANSAC, pose-only LM)
             over consecutive frames of one synthetic sequence (configs[2])
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL, used ONLY for the
barrier and the max-over-ranks clock); pairs are sharded round-robin (pair k -> rank
k mod N), no data-path collective, weak scaling (B pairs per rank per step).
Prints ONE JSON line on rank 0.
"""
import argparse
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
FAST_BYTES_PER_IMAGE = S_PYR * W * H  # the "S [FAST read]" term of that model


def make_pairs(B, rank, world):
    """B synthetic stereo pairs for this rank (pair k of the global stream -> rank k mod N)."""
    import util
    Ls, Rs = [], []
    for i in range(B):
        k = i * world + rank
        if k % 16 == 0:                      # every 16th pair: the real street scene
            L, R = util.urban_pair(W, H, x0=40 + (k // 16) % 60, y0=4 + (k // 16) % 10)
        else:
            L, R = util.shifted_pair(0x5EED0000 + k, W, H, disparity=4 + k % 40)
        Ls.append(L); Rs.append(R)
    return np.stack(Ls), np.stack(Rs)


def to_device(imgs, dev):
    t = torch.zeros((imgs.shape[0], H, PITCH), dtype=torch.uint8, device=dev)
    t[:, :, :W] = torch.from_numpy(imgs).to(dev)
    return t


def cpu_baseline(pairs_L, pairs_R, cam, budget_s=15.0):
    """The oracle (a single-threaded C port of the same path) timed on this box's host
    cores over a bounded sample of the same workload."""
    from oracle import binding as orc
    orc.build()
    n, t0 = 0, time.perf_counter()
    while n < len(pairs_L) and (time.perf_counter() - t0) < budget_s:
        orc.stereo_frame(pairs_L[n], pairs_R[n], cam.bf, cam.fx)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "stereo pairs/s", "cores": 1, "kind": "port",
            "sample": "%d pairs of the benchmark batch (ORB on L and R + sparse stereo), %.1f s, "
                      "oracle/libsvo_oracle.so single thread; host has %d cores"
                      % (n, dt, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="stereo pairs per step per GPU")
    ap.add_argument("--workload", default="frontend", choices=["frontend", "track"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    pkg = svo_loader.load()
    cam = pkg.Camera(**pkg.KITTI_00_02)
    B = args.batch
    svo = pkg.Svo(W, H, device=local, max_kp=500, max_batch=B)
    Lh, Rh = make_pairs(B, rank, world)
    dL, dR = to_device(Lh, dev), to_device(Rh, dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_depth = torch.zeros((B, 500), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        svo.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), PITCH, B, cam,
                               d_nL=d_n.data_ptr(), d_depth=d_depth.data_ptr())

    def fence():
        svo.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_profile:
        svo.profile_reset()
        svo.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = {}
    if not args.no_profile:
        svo.profile_enable(False)
        prof = svo.profile()

    n_kp = d_n.cpu().numpy()
    n_depth = int((d_depth > 0).sum().item())
    if rank == 0:
        pairs = world * B * args.steps
        out = {
            "metric": "stereo frames/sec on KITTI 00 (tracking front end)",
            "value": pairs / dt, "unit": "stereo pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "kitti00-shaped stereo pairs 1241x376: ORB pyramid (8 levels, 500 kp) on "
                                   "L and R + sparse epipolar stereo, HBM-resident (BASELINE configs[1], batched)",
                       "pairs_per_step_per_gpu": B, "sharding": "pair k -> rank k mod N, no collective",
                       "mean_keypoints_left": float(n_kp.mean()), "mean_stereo_depths": n_depth / B},
        }
        if prof:
            kern = {k: {"avg_ms": v[0] / max(v[1], 1), "launches": v[1]} for k, v in prof.items()}
            dom = max(prof.items(), key=lambda kv: kv[1][0])[0]
            dom_s = prof[dom][0] / max(prof[dom][1], 1) * 1e-3
            algo = {"k_fast": FAST_BYTES_PER_IMAGE * 2 * B,
                    "k_pyr_level": (2 * S_PYR - 1) * W * H * 2 * B}.get(dom, ALGO_BYTES_PER_PAIR * B)
            ach = algo / dom_s / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "algorithmic_bytes_per_launch": algo,
                               "pipeline_frac": ALGO_BYTES_PER_PAIR * (pairs / dt / world) / 1e9 / HBM_PEAK_GBS}
            out["kernels"] = kern
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Lh, Rh, cam)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    svo.close()


if __name__ == "__main__":
    main()
