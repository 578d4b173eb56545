"""The bench's tail legs in one process, in the order given (argv[1]: letters p = pnp_solver_modes, s = sharded, m = multi_sequence,
e = semantic_elas; may repeat), with the stream picker's diagnostics; SVO_QH_HW_QUEUES=<n> sets GPU_MAX_HW_QUEUES (default: the
runtime's own).  One JSON line per leg."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SVO_QH_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["SVO_QH_HW_QUEUES"]
import bench, svo_loader, torch  # noqa: E402
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
dL, dR, _ = bench.render_frames(synth, 1280, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
refn = bench.tail_leg_reference(pkg, cam, dL, dR, dev, 0, 1024)
names = {"p": "pnp_solver_modes", "s": "sharded", "m": "multi_sequence", "e": "semantic_elas"}
rec = pkg.TRACK_DTYPE.itemsize
fb = bench.H * bench.PITCH
for c in (sys.argv[1] if len(sys.argv) > 1 else "psme"):
    if c == "w":   # warm the runtime's hardware-queue pools: four used streams per priority, released again
        ss = [torch.cuda.Stream(device=dev, priority=p) for p in (-1, -1, -1, -1, 0, 0, 0, 0)]
        for st in ss:
            with torch.cuda.stream(st):
                torch.zeros(8, device=dev).add_(1)
        torch.cuda.synchronize()
        del ss
        print(json.dumps({"leg": "hardware-queue pools warmed (4 high + 4 normal priority streams used and released)"}), flush=True)
        continue
    if c in "hf":   # h: a headline-like context (batched tracker over 1024 frames), closed again; f: the same with the statistical solver + kernel times
        s = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=256)
        if c == "f":
            s.set_option("epnp_exact", 0)
            s.profile_reset(); s.profile_enable(True)
        s.track_reset(cam)
        r = torch.zeros((1024, rec), dtype=torch.uint8, device=dev)
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c0 in range(0, 1024, 128):
            s.track_batch_dev(dL.data_ptr() + c0 * fb, dR.data_ptr() + c0 * fb, bench.PITCH, 128, r.data_ptr() + c0 * rec)
        s.sync(); dt = time.perf_counter() - t0
        kern = None
        if c == "f":
            s.profile_enable(False)
            kern = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in s.profile().items()}
        print(json.dumps({"leg": "batched tracker, 1024 frames" + (" (statistical solver, timers on)" if c == "f" else ""), "value": 1024 / dt,
                          "stream_probe": list(s.debug_stream_probe()), "kernel_avg_us": kern}), flush=True)
        s.close()
        continue
    r = bench.run_tail_leg(names[c], pkg, cam, dL, dR, dev, 0, refn)
    print(json.dumps({"leg": names[c], "value": bench.leg_value(r), "stream_probe": r.get("stream_probe") if isinstance(r, dict) else None,
                      "error": r.get("error") if isinstance(r, dict) else None}), flush=True)
