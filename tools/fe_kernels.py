"""Front-end kernels A/B: for every option set given ("key=value,key=value" per argument; "" = defaults) the per-kernel HIP-event times of
the batched front end on ONE stream (64 pairs per call, synth-kitti 1241 x 376) and the rate of the default schedule (two slices)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, svo_loader, torch  # noqa: E402
pkg = svo_loader.load()
synth = importlib.import_module("stereo_semantic_vo_amd.synth")
dev = torch.device("cuda", 0)
B = 128
dL, dR, _ = bench.render_frames(synth, B, dev, synth.BASE_SEED)
cam = pkg.Camera(**pkg.KITTI_00_02)
K = 500
kp = torch.zeros((B, K, pkg.KP_DTYPE.itemsize), dtype=torch.uint8, device=dev); de = torch.zeros((B, K, 32), dtype=torch.uint8, device=dev)
n = torch.zeros(B, dtype=torch.int32, device=dev); dp = torch.zeros((B, K), dtype=torch.float32, device=dev)
ref = None
for opts in (sys.argv[1:] or [""]):
    s = pkg.Svo(bench.W, bench.H, device=0, max_kp=500, max_batch=B)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        s.set_option(k, int(v))
    def run():
        s.frontend_batch_dev(dL.data_ptr(), dR.data_ptr(), bench.PITCH, B, cam, d_kpL=kp.data_ptr(), d_descL=de.data_ptr(), d_nL=n.data_ptr(), d_depth=dp.data_ptr())
    for _ in range(3):
        run()
    s.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    s.sync()
    rate = 20 * B / (time.perf_counter() - t0)
    got = (kp.cpu().numpy().tobytes(), de.cpu().numpy().tobytes(), n.cpu().numpy().tobytes(), dp.cpu().numpy().tobytes())
    s.set_option("frontend_overlap", 0)
    s.profile_reset(); s.profile_enable(True)
    for _ in range(5):
        run()
    s.sync(); s.profile_enable(False)
    kern = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in s.profile().items()}
    s.close()
    same = None if ref is None else got == ref
    ref = ref or got
    print(json.dumps({"options": opts, "pairs_per_s": round(rate), "kernel_us_per_128_pairs_one_stream": kern, "sum_us": round(sum(kern.values()), 1),
                      "results_equal_first": same}), flush=True)
