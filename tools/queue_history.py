"""Does the tail's rate depend on what the process did with streams BEFORE the tracker context was made?

One process, the same 512 frames every time; the argument is a string of steps carried out in order:
  M  measure: svo_track_batch_dev on a NEW context (best of three calls): one JSON line - frames/s, the stream probe's outcome
     (candidates tried, two chains together / one alone in per cent), records equal to the first measurement's
  B  history: three idle high-priority streams are created and a pair of contexts that ran a sharded call is closed
  C  history: a many-sequence context (64 sequences) runs and is closed
  S  history: three contexts are created and stay open, idle (three high-priority streams)
  E  history: one context with room for 128 pairs is created and stays open, idle
  X  history: every context kept open by the steps before is closed
  D  history: the batched front end (frontend_batch_dev, several streams) runs on another context that STAYS open
  H  history: three idle high-priority streams made with hipStreamCreateWithPriority through torch + a closed tracker context
The process runs with the HIP runtime's default number of hardware queues per priority (4) unless SVO_QH_HW_QUEUES=<n> is set.
usage: python tools/queue_history.py [steps]      default MBMCMDM ("M" alone = the fresh-process figure)
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
if os.environ.get("SVO_QH_HW_QUEUES", "default") == "default":
    os.environ.pop("GPU_MAX_HW_QUEUES", None)   # (bench.py raises it to 16 for its own legs; a deployment runs with the runtime's default of 4)
else:
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["SVO_QH_HW_QUEUES"]
import svo_loader  # noqa: E402
import torch  # noqa: E402

pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
N = 512
dL, dR, _ = bench.render_frames(synth, N, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
rec = pkg.TRACK_DTYPE.itemsize
torch.cuda.synchronize()
keep = []


def tail_rate(tag, ref=None):
    s = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=N)
    for kv in filter(None, os.environ.get("SVO_BENCH_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        s.set_option(k, int(v))
    res = torch.zeros((N, rec), dtype=torch.uint8, device=dev)
    best = 0.0
    for rep in range(4):
        s.track_reset(cam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.track_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, N, res.data_ptr())
        s.sync()
        dt = time.perf_counter() - t0
        if rep:
            best = max(best, N / dt)
    kern = None
    if os.environ.get("SVO_QH_PROFILE"):
        s.profile_reset(); s.profile_enable(True)
        s.track_reset(cam)
        s.track_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, N, res.data_ptr())
        s.sync()
        s.profile_enable(False)
        kern = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in s.profile().items()}
    attempts, polls = s.debug_stream_probe()
    got = res.cpu().numpy().tobytes()
    s.close()
    print(json.dumps({"case": tag, "frames_per_s": round(best, 1), "probe_candidates": attempts, "probe_two_chains_vs_one_percent": polls,
                      "records_equal_first": None if ref is None else got == ref, "kernel_avg_us": kern}), flush=True)
    return got


ref = None
label = "fresh process"
for c in (sys.argv[1] if len(sys.argv) > 1 else "MBMCMDM"):
    if c == "M":
        got = tail_rate(label, ref)
        ref = ref or got
    elif c == "B":
        keep.extend(torch.cuda.Stream(device=dev, priority=-1) for _ in range(3))
        ctxs = [pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=64) for _ in range(2)]
        ctxs[0].track_reset(cam)
        r = torch.zeros((128, rec), dtype=torch.uint8, device=dev)
        Ls = [dL[g:128:2].contiguous() for g in range(2)]; Rs = [dR[g:128:2].contiguous() for g in range(2)]
        torch.cuda.synchronize()
        pkg.Svo.track_sharded_dev(ctxs, [t.data_ptr() for t in Ls], [t.data_ptr() for t in Rs], bench.PITCH, 128, r.data_ptr())
        ctxs[0].sync()
        for x in ctxs:
            x.close()
        label = "after 3 idle high-priority streams + a closed sharded pair"
    elif c == "C":
        m = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=64)
        m.track_multi_reset(64, cam)
        r = torch.zeros((64 * 4, rec), dtype=torch.uint8, device=dev)
        fb = bench.H * bench.PITCH
        for t in range(4):
            m.track_multi_step_dev(dL.data_ptr() + t * fb, dR.data_ptr() + t * fb, bench.PITCH, 64, r.data_ptr() + t * 64 * rec)
        m.sync(); m.close()
        label = "after a closed 64-sequence context"
    elif c == "D":
        f = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=128)
        K = 500
        kp = torch.zeros((128, K, pkg.KP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        de = torch.zeros((128, K, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(128, dtype=torch.int32, device=dev); dp = torch.zeros((128, K), dtype=torch.float32, device=dev)
        for _ in range(3):
            f.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, 128, cam, d_kpL=kp.data_ptr(), d_descL=de.data_ptr(),
                                 d_nL=n.data_ptr(), d_depth=dp.data_ptr())
        f.sync()
        keep.append(f)
        label = "beside an open context that ran the batched front end"
    elif c == "S":
        keep.extend(pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=1) for _ in range(3))
        label = "beside three open, idle contexts (three high-priority streams)"
    elif c == "E":
        keep.append(pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=128))
        label = "beside one open, idle context with room for 128 pairs"
    elif c == "X":
        for k in keep:
            if hasattr(k, "close"):
                k.close()
        keep.clear()
        label = "after the kept contexts were closed"
    elif c in "FGT":
        f = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=128)
        if c == "F":
            f.set_option("frontend_overlap", 0)      # the whole batch on the context's own (high-priority) stream
        if c == "T":
            st = torch.cuda.Stream(device=dev, priority=0)
            with torch.cuda.stream(st):
                torch.zeros(8, device=dev).add_(1)
            keep.append(st)
        else:
            K = 500
            kp = torch.zeros((128, K, pkg.KP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
            de = torch.zeros((128, K, 32), dtype=torch.uint8, device=dev)
            n = torch.zeros(128, dtype=torch.int32, device=dev); dp = torch.zeros((128, K), dtype=torch.float32, device=dev)
            nb = 128 if c == "F" else 2
            f.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, nb, cam, d_kpL=kp.data_ptr(), d_descL=de.data_ptr(),
                                 d_nL=n.data_ptr(), d_depth=dp.data_ptr())
            f.sync()
        keep.append(f)
        label = {"F": "beside an open context that ran the front end on its own stream only", "G": "beside an open context that ran a 2-pair front end",
                 "T": "beside an open idle context + a used normal-priority torch stream"}[c]
    elif c == "H":
        for _ in range(3):
            st = torch.cuda.Stream(device=dev, priority=-1)
            with torch.cuda.stream(st):
                torch.zeros(8, device=dev).add_(1)
            keep.append(st)
        torch.cuda.synchronize()
        x = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=16)
        x.track_reset(cam)
        r = torch.zeros((16, rec), dtype=torch.uint8, device=dev)
        x.track_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, 16, r.data_ptr())
        x.sync(); x.close()
        label = "after three idle high-priority streams and a closed tracker context"
