"""svo_elas_filter_lattice (host side of the library, no GPU needed): the vectorised clean-up of the support-point lattice
against a plain restatement of the reference's three in-place passes (Thirdparty/libelas/src/elas.cpp:152-265:
removeInconsistentSupportPoints, removeRedundantSupportPoints along columns, then along rows), on random lattices -
sparse, dense, noisy, with borders - and on parameter sets other than the defaults."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return C.CDLL(os.path.join(ROOT, "stereo-semantic-vo_amd", "libsvo_hip.so"))


def reference_passes(D, win, thr, min_support, red_dist, red_thr):
    D = D.copy()
    Hc, Wc = D.shape
    for u in range(Wc):                              # removeInconsistentSupportPoints (elas.cpp:152-196)
        for v in range(Hc):
            d = int(D[v, u])
            if d < 0:
                continue
            w = D[max(v - win, 0):min(v + win, Hc - 1) + 1, max(u - win, 0):min(u + win, Wc - 1) + 1].astype(np.int32)
            if int(((w >= 0) & (np.abs(w - d) <= thr)).sum()) < min_support:
                D[v, u] = -1
    for vertical in (True, False):                   # removeRedundantSupportPoints (elas.cpp:198-265)
        for u in range(Wc):
            for v in range(Hc):
                d = int(D[v, u])
                if d < 0:
                    continue
                redundant = True
                for sgn in (-1, 1):
                    support = False
                    for j in range(1, red_dist + 1):
                        u2, v2 = (u, v + sgn * j) if vertical else (u + sgn * j, v)
                        if u2 < 0 or v2 < 0 or u2 >= Wc or v2 >= Hc:
                            break
                        d2 = int(D[v2, u2])
                        if d2 >= 0 and abs(d - d2) <= red_thr:
                            support = True
                            break
                    if not support:
                        redundant = False
                        break
                if redundant:
                    D[v, u] = -1
    return D


@pytest.mark.parametrize("seed", range(24))
def test_vector_filter_equals_reference_passes(lib, seed):
    rng = np.random.default_rng(seed)
    Wc, Hc = int(rng.integers(1, 70)), int(rng.integers(1, 40))
    defaults = seed % 3 != 0
    win = 5 if defaults else int(rng.integers(0, 6))
    thr = 5 if defaults else int(rng.integers(0, 9))
    min_support = 5 if defaults else int(rng.integers(1, 12))
    red_dist = 5 if defaults else int(rng.integers(0, 6))
    red_thr = 1 if defaults else int(rng.integers(0, 4))
    v = np.arange(Hc)[:, None]
    D = (5 + (v * 40) // max(Hc, 1) + rng.integers(0, 3, (Hc, Wc))).astype(np.int16)
    D[rng.random((Hc, Wc)) < rng.random() * 0.9] = -1                          # holes
    noisy = rng.random((Hc, Wc)) < rng.random() * 0.3
    D[noisy] = rng.integers(0, 256, int(noisy.sum()))                           # outliers
    D[0, :] = 0; D[:, 0] = 0                                                    # lattice row / column 0 as the kernel leaves them
    want = reference_passes(D, win, thr, min_support, red_dist, red_thr)
    got = np.ascontiguousarray(D)
    rc = lib.svo_elas_filter_lattice(got.ctypes.data_as(C.c_void_p), Wc, Hc, win, thr, min_support, red_dist, red_thr)
    assert rc == 0
    assert np.array_equal(got, want)


def test_vector_filter_declines_what_it_does_not_cover(lib):
    D = np.zeros((8, 8), np.int16)
    assert lib.svo_elas_filter_lattice(D.ctypes.data_as(C.c_void_p), 8, 8, 6, 5, 5, 5, 1) != 0     # window > 5: the caller's scalar loops
    assert lib.svo_elas_filter_lattice(D.ctypes.data_as(C.c_void_p), 8, 8, 5, 5, 5, 7, 1) != 0
