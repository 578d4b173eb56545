"""Dense ELAS stereo on the GPU (svo_elas_process, through the C-ABI) against the REAL reference:
the compiled libelas of /root/reference (oracle/_ref/libref_elas.so, built by oracle/Makefile.ref; the
.so travels to the GPU box, /root/reference does not).  Every stage is compared bit for bit:
descriptors, support points, triangles, plane parameters, grids, raw disparity maps, and the maps after
the L/R check, segment removal, gap interpolation, adaptive mean and median.  Downstream of the
triangulation the reference's stages are fed the product's (canonically ordered) triangle lists - see
tests/test_elas_delaunay.py for why order is the one thing Triangle does not define."""
import numpy as np
import pytest

import util
from oracle import binding as ob
import svo_loader

svo = svo_loader.load()

pytestmark = [pytest.mark.gpu]
needs_ref = pytest.mark.skipif(ob.ref_elas_lib() is None, reason="oracle/_ref not built")

def cones_pair():
    """Middlebury `cones` at 900x750 (data files the reference ships under Thirdparty/libelas/img/):
    disparities up to ~220 px, so the whole 0..255 search range is exercised."""
    import os
    return (util.read_pgm(os.path.join(util.GOLDEN, "cones_left.pgm")),
            util.read_pgm(os.path.join(util.GOLDEN, "cones_right.pgm")))


PAIRS = {
    "urban-kitti-crop": lambda: util.urban_pair(),
    "urban-small": lambda: util.urban_pair(640, 240, 300, 60),
    "urban-odd-size": lambda: util.urban_pair(777, 301, 111, 33),
    "cones": cones_pair,
    "tiny": lambda: util.urban_pair(97, 64, 600, 150),
}
STAGES = ("desc1", "desc2", "support", "planes1", "planes2", "grid1", "grid2",
          "D1_raw", "D2_raw", "D1_lr", "D2_lr", "D1_seg", "D2_seg", "D1_gap", "D2_gap",
          "D1_mean", "D2_mean", "D1", "D2")


def same_triangulation(mine, ref, xy):
    """Identical index triples; with duplicate points (which twin Triangle keeps is an artefact of its
    randomised quicksort) identical geometry."""
    ref = ob.canonical_triangles(ref)
    if len(np.unique(xy, axis=0)) == len(xy):
        return mine.shape == ref.shape and np.array_equal(mine, ref)
    geo = lambda t: set(tuple(sorted(map(tuple, xy[list(r)]))) for r in t)
    return geo(mine) == geo(ref)


def params(middlebury):
    a = svo.elas_default_params(1 if middlebury else 0)
    b = ob.ref_elas_params(middlebury)
    for f, _ in a._fields_:
        assert getattr(a, f) == getattr(b, f), f      # same defaults as Elas::parameters(setting)
    return a, b


@pytest.fixture(scope="module")
def ctx():
    c = svo.Svo(util.KITTI_W, util.KITTI_H)
    yield c
    c.close()


@needs_ref
@pytest.mark.parametrize("middlebury", [False, True])
@pytest.mark.parametrize("pair", sorted(PAIRS))
def test_every_stage_bit_exact(ctx, pair, middlebury):
    L, R = PAIRS[pair]()
    pa, pb = params(middlebury)
    g = ctx.elas_process(L, R, pa, taps=True)
    r = ob.ref_elas_staged(L, R, pb, tri1=g["tri1"], tri2=g["tri2"])
    assert len(g["support"]) > (100 if pair != "tiny" else 3)
    for k in STAGES:
        assert g[k].shape == r[k].shape, k
        assert np.array_equal(g[k], r[k]), (k, int((g[k] != r[k]).sum()))
    # the triangulation itself: identical to Triangle's up to order
    r0 = ob.ref_elas_staged(L, R, pb)
    sp = g["support"]
    assert same_triangulation(g["tri1"], r0["tri1"], sp[:, :2])
    assert same_triangulation(g["tri2"], r0["tri2"], np.stack([sp[:, 0] - sp[:, 2], sp[:, 1]], 1))
    # end to end against the untouched Elas::process: only pixels on shared triangle edges may differ
    assert (g["D1"] != r0["D1"]).mean() < 2e-3 and (g["D2"] != r0["D2"]).mean() < 2e-3


CUSTOM = [
    dict(disp_max=63, grid_size=16, candidate_stepsize=4, incon_window_size=3, incon_min_support=3),
    dict(sigma=2.0, sradius=3.0, gamma=4.0, beta=0.03, match_texture=5, support_texture=20),
    dict(ipol_gap_width=7, speckle_size=100, speckle_sim_threshold=2.0, lr_threshold=1,
         filter_median=1, filter_adaptive_mean=1, postprocess_only_left=0),
    dict(disp_min=3, disp_max=200, support_threshold=0.9, add_corners=1, incon_threshold=3, grid_size=25),
]


@needs_ref
@pytest.mark.parametrize("custom", range(len(CUSTOM)))
def test_non_default_parameters_bit_exact(ctx, custom):
    L, R = PAIRS["urban-small"]()
    pa, pb = params(False)
    for k, v in CUSTOM[custom].items():
        setattr(pa, k, v); setattr(pb, k, v)
    g = ctx.elas_process(L, R, pa, taps=True)
    r = ob.ref_elas_staged(L, R, pb, tri1=g["tri1"], tri2=g["tri2"])
    assert len(g["support"]) > 50
    for k in STAGES:
        assert np.array_equal(g[k], r[k]), (k, int((g[k] != r[k]).sum()))


def test_plain_call_equals_tapped_call_and_is_deterministic(ctx):
    L, R = PAIRS["urban-small"]()
    D1, D2 = ctx.elas_process(L, R)
    g = ctx.elas_process(L, R, taps=True)
    E1, E2 = ctx.elas_process(L, R)
    assert np.array_equal(D1, g["D1"]) and np.array_equal(D2, g["D2"])
    assert np.array_equal(D1, E1) and np.array_equal(D2, E2)
    assert (D1 >= 0).mean() > 0.5


def test_textureless_pair_leaves_outputs_untouched(ctx):
    """< 3 support points: the reference returns without writing D1/D2 (elas.cpp:70-75)."""
    L = np.full((120, 200), 90, np.uint8)
    g = ctx.elas_process(L, L, taps=True)
    assert len(g["support"]) == 0 and (g["D1"] == 0).all() and (g["D2"] == 0).all()


@needs_ref
@pytest.mark.parametrize("middlebury", [False, True])
@pytest.mark.parametrize("pair", ["urban-small", "urban-odd-size"])
def test_subsampling_bit_exact(ctx, pair, middlebury):
    """param.subsampling = 1: every second pixel, maps of (W/2) x (H/2), lattice step 6, 4-tap adaptive
    mean, halved gap width, speckle size 2*sqrt(speckle_size) (elas.cpp:379-381, 846-871, 986-991)."""
    L, R = PAIRS[pair]()
    pa, pb = params(middlebury)
    pa.subsampling = 1; pb.subsampling = 1
    g = ctx.elas_process(L, R, pa, taps=True)
    assert g["D1"].shape == (L.shape[0] // 2, L.shape[1] // 2)
    r = ob.ref_elas_staged(L, R, pb, tri1=g["tri1"], tri2=g["tri2"])
    assert len(g["support"]) > 50
    for k in STAGES:
        assert g[k].shape == r[k].shape, k
        assert np.array_equal(g[k], r[k]), (k, int((g[k] != r[k]).sum()))
    D1, D2 = ctx.elas_process(L, R, pa)
    assert np.array_equal(D1, g["D1"]) and np.array_equal(D2, g["D2"])


def test_rejects_bad_parameters(ctx):
    L, R = PAIRS["urban-small"]()
    for field, value in (("subsampling", 2), ("disp_max", 300), ("grid_size", 0)):
        p = svo.elas_default_params(0)
        setattr(p, field, value)
        with pytest.raises(svo.SvoError):
            ctx.elas_process(L, R, p)


# ---- committed golden vectors (tests/golden/elas_vectors.npz, made by tests/golden/make_elas_golden.py from the
# ---- compiled reference): this test needs no oracle/_ref at run time
GOLD = np.load(__import__("os").path.join(util.GOLDEN, "elas_vectors.npz"))


@pytest.mark.parametrize("name", ["small", "tiny"])
@pytest.mark.parametrize("middlebury", [0, 1])
def test_against_committed_reference_vectors(ctx, name, middlebury):
    k = "%s_%d_" % (name, middlebury)
    L, R = util.urban_pair(*[int(v) for v in GOLD[k + "crop"]])
    g = ctx.elas_process(L, R, svo.elas_default_params(middlebury), taps=True)
    assert np.array_equal(g["support"], GOLD[k + "support"])
    assert np.array_equal(g["tri1"], GOLD[k + "tri1"]) and np.array_equal(g["tri2"], GOLD[k + "tri2"])
    assert np.array_equal(g["D1"], GOLD[k + "D1"]) and np.array_equal(g["D2"], GOLD[k + "D2"])
    # the untouched Elas::process differs from that only where Triangle's triangle ORDER matters
    for side in ("1", "2"):
        exp = GOLD[k + "D" + side].ravel().copy()
        exp[GOLD[k + "process_diff_idx" + side]] = GOLD[k + "process_diff_val" + side]
        assert (exp != g["D" + side].ravel()).sum() == len(GOLD[k + "process_diff_idx" + side]) <= 2


@pytest.mark.parametrize("middlebury", [0, 1])
def test_batch_equals_one_pair_at_a_time(ctx, middlebury):
    """svo_elas_batch_dev (pairs and maps resident in HBM, streams + host thread pool) == svo_elas_process per
    pair, bit for bit; a textureless pair in the batch is reported as not produced and left untouched."""
    import torch
    W, H = 640, 240
    crops = [(300, 60), (100, 20), (500, 100), (640, 140), (0, 0), (333, 77), (10, 150), (700, 30), (250, 110)]
    pairs = [util.urban_pair(W, H, x, y) for x, y in crops]
    flat = np.full((H, W), 77, np.uint8)
    pairs.insert(4, (flat, flat))
    B = len(pairs)
    p = svo.elas_default_params(middlebury)
    dev = torch.device("cuda", 0)
    stride = 704
    dL = torch.zeros((B, H, stride), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    for b, (L, R) in enumerate(pairs):
        dL[b, :, :W] = torch.from_numpy(L).to(dev); dR[b, :, :W] = torch.from_numpy(R).to(dev)
    D1 = torch.full((B, H, W), 123.0, dtype=torch.float32, device=dev); D2 = torch.full_like(D1, 123.0)
    torch.cuda.synchronize()
    for rep in range(2):     # second call reuses the per-pair states
        produced = ctx.elas_batch_dev(dL.data_ptr(), dR.data_ptr(), stride, W, H, B, D1.data_ptr(), D2.data_ptr(), p)
    h1, h2 = D1.cpu().numpy(), D2.cpu().numpy()
    # with add_corners (MIDDLEBURY) the six corner points alone are a triangulation: the reference goes on
    assert produced.tolist() == [1, 1, 1, 1, 1 if middlebury else 0, 1, 1, 1, 1, 1]
    for b, (L, R) in enumerate(pairs):
        if not produced[b]:
            assert (h1[b] == 123.0).all() and (h2[b] == 123.0).all()
            continue
        e1, e2 = ctx.elas_process(L, R, p)
        assert np.array_equal(h1[b], e1) and np.array_equal(h2[b], e2), b


def test_batch_argument_checks(ctx):
    import torch
    dev = torch.device("cuda", 0)
    z = torch.zeros((2, 64, 128), dtype=torch.uint8, device=dev)
    D = torch.zeros((2, 64, 128), dtype=torch.float32, device=dev)
    with pytest.raises(svo.SvoError):      # B < 1
        ctx.elas_batch_dev(z.data_ptr(), z.data_ptr(), 128, 128, 64, 0, D.data_ptr(), D.data_ptr())
    with pytest.raises(svo.SvoError):      # stride < width
        ctx.elas_batch_dev(z.data_ptr(), z.data_ptr(), 64, 128, 64, 2, D.data_ptr(), D.data_ptr())
    with pytest.raises(svo.SvoError):      # null output
        ctx.elas_batch_dev(z.data_ptr(), z.data_ptr(), 128, 128, 64, 2, 0, D.data_ptr())
    produced = ctx.elas_batch_dev(z.data_ptr(), z.data_ptr(), 128, 128, 64, 2, D.data_ptr(), D.data_ptr())
    assert produced.tolist() == [0, 0]     # black images: no support points, nothing written
