"""The pipelined host-fed entries (svo_track_batch_host, svo_track_sharded_host, svo_frontend_batch_host): the images of a
call start in HOST memory - main.cpp:159-195 reads one stereo pair from disk per Tracking::Track; SURVEY.md section 8e: "H2D
2 P bytes, D2H ~30 KB" per pair, pair k uploaded to GPU k mod G.  Whatever the source (pinned memory copied where it lies,
pinned memory of another pitch, pageable memory staged by worker threads), however the calls are cut and without a
synchronisation between them, the records must equal svo_track_batch_dev's on the same frames byte for byte."""
import importlib

import numpy as np
import pytest

N = 48
PITCH = 1280


@pytest.fixture(scope="module")
def seq(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    synth = importlib.import_module("stereo_semantic_vo_amd.synth")
    dev = torch.device("cuda", 0)
    L, R, _ = synth.render_sequence(N, device=dev)
    H, W = int(L.shape[1]), int(L.shape[2])
    dL = torch.zeros((N, H, PITCH), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    dL[:, :, :W] = L; dR[:, :, :W] = R
    cam = pkg.Camera(**pkg.KITTI_00_02)
    return dict(dL=dL, dR=dR, W=W, H=H, cam=cam, dev=dev,
                hL=dL.cpu().numpy(), hR=dR.cpu().numpy(),                       # pitch 1280 (= the library's staging pitch)
                pL=np.ascontiguousarray(L.cpu().numpy()), pR=np.ascontiguousarray(R.cpu().numpy()))   # packed, stride = W


def boxes_of(k):
    x = 80 + (4 * k) % 820
    return [[x, x + 260, 150, 330], [100, 260, 200, 300 + (2 * k) % 60]]


def reference(pkg, s, sizes, boxes=False, depth_source=0):
    """svo_track_batch_dev on the resident copies, cut into the same calls"""
    import torch
    ctx = pkg.Svo(s["W"], s["H"], max_batch=max(sizes))
    if depth_source:
        ctx.set_option("depth_source", depth_source)
    ctx.track_reset(s["cam"])
    rec = pkg.TRACK_DTYPE.itemsize
    n = sum(sizes)
    res = torch.zeros((n, rec), dtype=torch.uint8, device=s["dev"])
    keep = None
    if boxes:
        b = np.array([boxes_of(k) for k in range(n)], np.int32)
        tb = torch.from_numpy(b).to(s["dev"]); tn = torch.full((n,), 2, dtype=torch.int32, device=s["dev"])
        keep = (tb, tn)
    fb = s["H"] * PITCH
    k0 = 0
    for B in sizes:
        bx = pkg.boxes_dev(keep[0].data_ptr() + k0 * 2 * 16, keep[1].data_ptr() + k0 * 4, 2) if boxes else None
        ctx.track_batch_dev(s["dL"].data_ptr() + k0 * fb, s["dR"].data_ptr() + k0 * fb, PITCH, B, res.data_ptr() + k0 * rec, boxes=bx)
        k0 += B
    ctx.sync()
    assert ctx.track_overflowed() == 0
    out = res.cpu().numpy().tobytes()
    ctx.close()
    return out


def pin(a):
    import torch
    t = torch.from_numpy(a).pin_memory()
    assert t.is_pinned()
    return t


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["pinned", "pinned_packed", "pageable", "pageable_packed"])
def test_host_fed_calls_back_to_back_equal_the_resident_tracker(pkg, seq, source):
    """Five calls of uneven sizes, no synchronisation in between; the records are read after ONE svo_sync."""
    s = seq
    sizes = [9, 17, 1, 13, 8]
    assert sum(sizes) == N
    want = reference(pkg, s, sizes)
    packed = source.endswith("packed")
    aL, aR = (s["pL"], s["pR"]) if packed else (s["hL"], s["hR"])
    stride = s["W"] if packed else PITCH
    keep = None
    if source.startswith("pinned"):
        keep = (pin(aL), pin(aR))
        pl, pr = keep[0].data_ptr(), keep[1].data_ptr()
    else:
        pl, pr = aL.ctypes.data, aR.ctypes.data
    ctx = pkg.Svo(s["W"], s["H"], max_batch=max(sizes))
    ctx.track_reset(s["cam"])
    res = np.zeros(N, pkg.TRACK_DTYPE)
    res["frame_id"] = -7
    fb = s["H"] * stride
    k0 = 0
    for B in sizes:
        ctx.track_batch_host(pl + k0 * fb, pr + k0 * fb, stride, B, res[k0:k0 + B])
        k0 += B
    ctx.sync()
    assert ctx.track_overflowed() == 0
    ctx.close()
    assert list(res["frame_id"]) == list(range(N))
    assert res.tobytes() == want


@pytest.mark.gpu
def test_host_fed_results_into_pinned_memory_and_a_second_sequence_on_the_same_context(pkg, seq):
    """Records written straight into a pinned result array; after svo_track_reset the same context tracks the sequence again
    (both image sets and both halves of everything are reused) with the same outcome."""
    import torch
    s = seq
    sizes = [16, 16, 16]
    want = reference(pkg, s, sizes)
    tL, tR = pin(s["hL"]), pin(s["hR"])
    ctx = pkg.Svo(s["W"], s["H"], max_batch=16)
    rec = pkg.TRACK_DTYPE.itemsize
    fb = s["H"] * PITCH
    for rep in range(2):
        res = torch.zeros((N, rec), dtype=torch.uint8).pin_memory()
        ctx.track_reset(s["cam"])
        for c in range(3):
            ctx.track_batch_host(tL.data_ptr() + c * 16 * fb, tR.data_ptr() + c * 16 * fb, PITCH, 16, res.data_ptr() + c * 16 * rec)
        ctx.sync()
        assert res.numpy().tobytes() == want, rep
    ctx.close()


@pytest.mark.gpu
def test_host_fed_with_detection_boxes(pkg, seq):
    s = seq
    sizes = [20, 28]
    want = reference(pkg, s, sizes, boxes=True)
    ctx = pkg.Svo(s["W"], s["H"], max_batch=max(sizes))
    ctx.track_reset(s["cam"])
    res = np.zeros(N, pkg.TRACK_DTYPE)
    fb = s["H"] * PITCH
    k0 = 0
    keep = []
    for B in sizes:
        bx = pkg.boxes_host(np.array([boxes_of(k) for k in range(k0, k0 + B)], np.int32), np.full(B, 2, np.int32))
        keep.append(bx)
        ctx.track_batch_host(s["hL"].ctypes.data + k0 * fb, s["hR"].ctypes.data + k0 * fb, PITCH, B, res[k0:k0 + B], boxes=bx)
        k0 += B
    ctx.sync()
    ctx.close()
    assert res.tobytes() == want


@pytest.mark.gpu
def test_host_fed_with_the_dense_elas_depth_source(pkg, seq):
    """configs[4]'s data flow host-fed: the dense stage waits for the call's uploads as a whole."""
    s = seq
    sizes = [12, 12]
    want = reference(pkg, s, sizes, boxes=True, depth_source=1)
    ctx = pkg.Svo(s["W"], s["H"], max_batch=12)
    ctx.set_option("depth_source", 1)
    ctx.track_reset(s["cam"])
    res = np.zeros(24, pkg.TRACK_DTYPE)
    fb = s["H"] * PITCH
    keep = []
    for c in range(2):
        bx = pkg.boxes_host(np.array([boxes_of(k) for k in range(c * 12, c * 12 + 12)], np.int32), np.full(12, 2, np.int32))
        keep.append(bx)
        ctx.track_batch_host(s["hL"].ctypes.data + c * 12 * fb, s["hR"].ctypes.data + c * 12 * fb, PITCH, 12, res[c * 12:c * 12 + 12], boxes=bx)
    ctx.sync()
    ctx.close()
    assert res.tobytes() == want


@pytest.mark.gpu
@pytest.mark.parametrize("G,source", [(2, "pinned"), (3, "pageable_packed")])
def test_sharded_host_uploads_pair_k_to_context_k_mod_G(pkg, seq, G, source):
    s = seq
    sizes = [22, 5, 21]
    want = reference(pkg, s, sizes)
    packed = source.endswith("packed")
    aL, aR = (s["pL"], s["pR"]) if packed else (s["hL"], s["hR"])
    stride = s["W"] if packed else PITCH
    keep = None
    if source.startswith("pinned"):
        keep = (pin(aL), pin(aR))
        pl, pr = keep[0].data_ptr(), keep[1].data_ptr()
    else:
        pl, pr = aL.ctypes.data, aR.ctypes.data
    ctxs = [pkg.Svo(s["W"], s["H"], max_batch=(max(sizes) + G - 1) // G) for _ in range(G)]
    ctxs[0].track_reset(s["cam"])
    res = np.zeros(N, pkg.TRACK_DTYPE)
    fb = s["H"] * stride
    k0 = 0
    for B in sizes:
        pkg.Svo.track_sharded_host(ctxs, pl + k0 * fb, pr + k0 * fb, stride, B, res[k0:k0 + B])
        k0 += B
    ctxs[0].sync()
    assert ctxs[0].track_overflowed() == 0
    for c in ctxs:
        c.close()
    assert res.tobytes() == want


@pytest.mark.gpu
def test_frontend_batch_host_equals_the_resident_front_end(pkg, seq):
    import torch
    s = seq
    K, B = 500, 16
    dev = s["dev"]
    ref = pkg.Svo(s["W"], s["H"], max_batch=B)
    kp = torch.zeros((N, K, pkg.KP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    desc = torch.zeros((N, K, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(N, dtype=torch.int32, device=dev)
    uR = torch.zeros((N, K), dtype=torch.float32, device=dev)
    depth = torch.zeros((N, K), dtype=torch.float32, device=dev)
    fb = s["H"] * PITCH
    for c in range(N // B):
        ref.frontend_batch_dev(s["dL"].data_ptr() + c * B * fb, s["dR"].data_ptr() + c * B * fb, PITCH, B, s["cam"],
                               d_kpL=kp[c * B:].data_ptr(), d_descL=desc[c * B:].data_ptr(), d_nL=n[c * B:].data_ptr(),
                               d_uR=uR[c * B:].data_ptr(), d_depth=depth[c * B:].data_ptr())
    ref.sync(); ref.close()
    ctx = pkg.Svo(s["W"], s["H"], max_batch=B)
    hkp = np.zeros((N, K), pkg.KP_DTYPE); hdesc = np.zeros((N, K, 32), np.uint8); hn = np.zeros(N, np.int32)
    huR = np.zeros((N, K), np.float32); hdepth = np.zeros((N, K), np.float32)
    fbp = s["H"] * s["W"]
    for c in range(N // B):      # packed pageable source, three calls back to back
        ctx.frontend_batch_host(s["pL"].ctypes.data + c * B * fbp, s["pR"].ctypes.data + c * B * fbp, s["W"], B, s["cam"],
                                kpL=hkp[c * B:], descL=hdesc[c * B:], nL=hn[c * B:], uR=huR[c * B:], depth=hdepth[c * B:])
    ctx.sync(); ctx.close()
    nn = n.cpu().numpy()
    assert np.array_equal(hn, nn) and nn.min() > 100
    gk = kp.cpu().numpy().view(pkg.KP_DTYPE).reshape(N, K)
    for f in range(N):
        m = int(nn[f])
        assert hkp[f, :m].tobytes() == gk[f, :m].tobytes(), f
        assert np.array_equal(hdesc[f, :m], desc[f, :m].cpu().numpy()), f
        assert np.array_equal(huR[f, :m].view(np.uint32), uR[f, :m].cpu().numpy().view(np.uint32)), f
        assert np.array_equal(hdepth[f, :m].view(np.uint32), depth[f, :m].cpu().numpy().view(np.uint32)), f


@pytest.mark.gpu
def test_stream_modes_per_context(pkg, seq):
    """svo_create_ex: the stream mode is a property of the context - a pooled and a dedicated context side by side
    in one process, both tracking the same frames to the same records; svo_stream_mode reports what is in effect."""
    s = seq
    sizes = [24, 24]
    want = reference(pkg, s, sizes)
    fb = s["H"] * PITCH
    for flags, mode in ((pkg.CREATE_POOLED_STREAMS, 0), (pkg.CREATE_TAIL_ALL_CUS, 1), (0, 3)):
        ctx = pkg.Svo(s["W"], s["H"], max_batch=24, flags=flags)
        assert ctx.stream_mode() == mode
        ctx.track_reset(s["cam"])
        res = np.zeros(N, pkg.TRACK_DTYPE)
        for c in range(2):
            ctx.track_batch_host(s["hL"].ctypes.data + c * 24 * fb, s["hR"].ctypes.data + c * 24 * fb, PITCH, 24, res[c * 24:c * 24 + 24])
        ctx.sync(); ctx.close()
        assert res.tobytes() == want, flags


@pytest.mark.gpu
def test_host_fed_argument_checks_and_capacity(pkg, seq):
    """Error behaviour of the host-fed entries: return codes, never a crash; a refused call leaves the context usable."""
    import ctypes as C
    s = seq
    ctx = pkg.Svo(s["W"], s["H"], max_batch=8)
    res = np.zeros(16, pkg.TRACK_DTYPE)
    L, R = s["hL"], s["hR"]
    lib = ctx.lib
    call = lambda l, r, stride, B, out: lib.svo_track_batch_host(ctx.h, l, r, int(stride), int(B), None, out)
    pl, pr, po = L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p)
    assert call(pl, pr, PITCH, 4, po) == -1                     # no svo_track_reset yet: invalid
    ctx.track_reset(s["cam"])
    assert call(None, pr, PITCH, 4, po) == -1 and call(pl, pr, PITCH, 4, None) == -1
    assert call(pl, pr, s["W"] - 1, 4, po) == -1                # stride below the width
    assert call(pl, pr, PITCH, 0, po) == -1
    assert call(pl, pr, PITCH, 9, po) == -5                     # more frames than max_batch: capacity
    bad = pkg.boxes_host(np.zeros((4, 2, 4), np.int32), np.array([2, 3, 0, 0], np.int32))     # n[1] exceeds the stride
    assert lib.svo_track_batch_host(ctx.h, pl, pr, PITCH, 4, C.byref(bad), po) == -1
    assert lib.svo_frontend_batch_host(ctx.h, pl, pr, PITCH, 9, C.byref(s["cam"]), None, None, None, None, None) == -5
    # the context still tracks
    want = reference(pkg, s, [8])
    ctx.track_batch_host(L.ctypes.data, R.ctypes.data, PITCH, 8, res[:8])
    ctx.sync()
    ctx.close()
    assert res[:8].tobytes() == want
