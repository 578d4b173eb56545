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
