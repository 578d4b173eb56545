"""MSA dense stereo, stages built so far: the ctmf median filter
(Thirdparty/MB/ctmf.c), pinned on the reference's own compiled ctmf (oracle/_ref/libref_ctmf.so).
CPU: the compiled reference equals the plain definition (true median, window clamped to the image).
GPU: svo_ctmf through the C-ABI equals the compiled reference, bit for bit, at the radii / channel counts
MSA uses (MSA.cpp:58-59: r = 1 on 3 channels; MSA.cpp:1006: r = 2 on 1 channel) and others."""
import os

import numpy as np
import pytest

import util
from oracle import binding as ob

REF = os.path.join(os.path.dirname(ob.__file__), "_ref", "libref_ctmf.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")

CASES = [((40, 57, 3), 1), ((33, 70), 2), ((64, 64, 3), 2), ((20, 31), 3), ((7, 6), 2), ((3, 9, 3), 1)]   # the reference asserts on images smaller than its window


def median_clamped(a, r):
    H, W = a.shape[:2]
    p = np.pad(a, ((r, r), (r, r)) + ((0, 0),) * (a.ndim - 2), mode="edge")
    st = np.stack([p[dy:dy + H, dx:dx + W] for dy in range(2 * r + 1) for dx in range(2 * r + 1)], 0)
    return np.sort(st, 0)[(2 * r + 1) ** 2 // 2]


@needs_ref
@pytest.mark.parametrize("shape,r", CASES)
def test_reference_ctmf_is_the_clamped_median(shape, r):
    a = np.random.default_rng(3).integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(ob.ref_ctmf(a, r), median_clamped(a, r))


@pytest.mark.gpu
def test_gpu_ctmf_equals_reference(pkg):
    """Against the compiled reference when oracle/_ref travelled with the snapshot, else against the plain
    definition the CPU test pins the reference to."""
    ref = ob.ref_ctmf if os.path.exists(REF) else median_clamped
    s = pkg.Svo(640, 240)
    rng = np.random.default_rng(5)
    for shape, r in CASES + [((376, 1241, 3), 1), ((376, 1241), 2)]:
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(s.ctmf(a, r), ref(a, r)), (shape, r)
    L, _ = util.urban_pair()
    bgr = np.stack([L, np.roll(L, 1, 1), np.roll(L, 2, 0)], 2)     # real image content, 3 channels
    assert np.array_equal(s.ctmf(bgr, 1), ref(bgr, 1))
    assert np.array_equal(s.ctmf(L, 2), ref(L, 2))
    with pytest.raises(pkg.SvoError):
        s.ctmf(L, 4)
    s.close()


# ---- MSA::init (gray, gradients, cost volumes, median images, gradients of those) ------------------------
def colour_pair(W=200, H=120):
    """A BGR stereo pair made of three differently shifted copies of the real urban1 crops."""
    L, R = util.urban_pair(W, H, 400, 100)
    mk = lambda g: np.ascontiguousarray(np.stack([g, np.roll(g, 1, 1), np.roll(g, 1, 0)], 2))
    return mk(L), mk(R)


def test_oracle_msa_init_against_numpy_twin():
    """oracle/orc_msa.c (the C restatement of MSA.cpp:22-139) against an independent float64 numpy twin of the same
    formulas, and its median images against the reference's compiled ctmf when that is present."""
    bl, br = colour_pair()
    o = ob.msa_init(bl, br, 49)
    gray = lambda b: (0.299 * b[..., 2] + 0.587 * b[..., 1] + 0.114 * b[..., 0] + 0.5).astype(np.int64).astype(np.uint8)

    def grad(img, axis, off):
        f = np.moveaxis(img.astype(np.float64), axis, 1)
        g = np.empty_like(f)
        g[:, 1:-1] = (f[:, 2:] - f[:, :-2]) * 0.5 + off
        g[:, 0] = f[:, 1] - f[:, 0] + off
        g[:, -1] = f[:, -1] - f[:, -2] + off
        return np.moveaxis(g, 1, axis)

    gL, gR = grad(gray(bl), 1, 127.5), grad(gray(br), 1, 127.5)
    W = bl.shape[1]
    for d in (0, 1, 7, 30, 48):
        idx = np.where(np.arange(W) - d >= 0, np.arange(W) - d, 0)
        dg = np.minimum(np.abs(gL - gR[:, idx]), 2.0)
        dc = np.minimum(np.abs(bl.astype(np.int64) - br[:, idx].astype(np.int64)).sum(2) / 3.0, 7.0)
        assert np.array_equal((0.11 * dc + (1 - 0.11) * dg).astype(np.float32), o["costL"][:, :, d])
        dd = np.minimum(d, W - 1 - np.arange(W))
        assert np.array_equal(o["costR"][:, :, d], o["costL"][:, np.arange(W) + dd, dd])
    assert np.array_equal(o["m3L"], median_clamped(bl, 1)) and np.array_equal(o["m3R"], median_clamped(br, 1))
    if os.path.exists(REF):
        assert np.array_equal(o["m3L"], ob.ref_ctmf(bl, 1))
    assert np.array_equal(o["r_graL"], grad(gray(o["m3L"]), 1, 0.0)) and np.array_equal(o["c_graR"], grad(gray(o["m3R"]), 0, 0.0))


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,disp", [(200, 120, 49), (97, 64, 17), (640, 240, 49)])
def test_gpu_msa_init_equals_oracle(pkg, W, H, disp):
    """svo_msa_init through the C-ABI == the oracle, bit for bit (float32 costs, float64 gradients, bytes)."""
    bl, br = colour_pair(W, H)
    s = pkg.Svo(640, 240)
    g = s.msa_init(bl, br, disp)
    r = ob.msa_init(bl, br, disp)
    for k in ("costL", "costR", "m3L", "m3R", "r_graL", "c_graL", "r_graR", "c_graR"):
        assert g[k].tobytes() == r[k].tobytes(), k
    s.close()
