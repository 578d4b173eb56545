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

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(ob.ref_elas_lib() is None, reason="oracle/_ref not built")]

PAIRS = {
    "urban-kitti-crop": lambda: util.urban_pair(),
    "urban-small": lambda: util.urban_pair(640, 240, 300, 60),
    "urban-odd-size": lambda: util.urban_pair(777, 301, 111, 33),
}


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


@pytest.mark.parametrize("middlebury", [False, True])
@pytest.mark.parametrize("pair", sorted(PAIRS))
def test_every_stage_bit_exact(ctx, pair, middlebury):
    L, R = PAIRS[pair]()
    pa, pb = params(middlebury)
    g = ctx.elas_process(L, R, pa, taps=True)
    r = ob.ref_elas_staged(L, R, pb, tri1=g["tri1"], tri2=g["tri2"])
    assert len(g["support"]) > 100
    for k in ("desc1", "desc2", "support", "planes1", "planes2", "grid1", "grid2",
              "D1_raw", "D2_raw", "D1_lr", "D2_lr", "D1_seg", "D2_seg", "D1_gap", "D2_gap",
              "D1_mean", "D2_mean", "D1", "D2"):
        assert g[k].shape == r[k].shape, k
        assert np.array_equal(g[k], r[k]), (k, int((g[k] != r[k]).sum()))
    # the triangulation itself: identical to Triangle's up to order
    r0 = ob.ref_elas_staged(L, R, pb)
    assert np.array_equal(g["tri1"], ob.canonical_triangles(r0["tri1"]))
    assert np.array_equal(g["tri2"], ob.canonical_triangles(r0["tri2"]))
    # end to end against the untouched Elas::process: only pixels on shared triangle edges may differ
    assert (g["D1"] != r0["D1"]).mean() < 2e-3 and (g["D2"] != r0["D2"]).mean() < 2e-3


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


def test_rejects_subsampling(ctx):
    p = svo.elas_default_params(0)
    p.subsampling = 1
    L, R = PAIRS["urban-small"]()
    with pytest.raises(svo.SvoError):
        ctx.elas_process(L, R, p)
