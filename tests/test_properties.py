"""Property tests (hypothesis) of the CPU oracle against brute-force definitions: the argmin /
runner-up quirk and its tie-breaking, the greedy claim order, the FAST corner score and NMS, the
fixed-point resize.  CPU only; the GPU suite then pins the HIP path to this oracle."""
import ctypes as C
import os
import sys

import numpy as np
from hypothesis import given, settings, strategies as st

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as twin  # noqa: E402  (independent numpy/pure-Python restatement)

SET = dict(max_examples=30, deadline=None)


def descs(rng, n, pool):
    """n descriptors drawn from a small pool of prototypes with a few flipped bits -> many ties."""
    protos = rng.integers(0, 256, (pool, 32), dtype=np.uint8)
    out = protos[rng.integers(0, pool, n)].copy()
    for i in range(n):
        for _ in range(int(rng.integers(0, 3))):
            b = int(rng.integers(0, 256))
            out[i, b // 8] ^= np.uint8(1 << (b % 8))
    return out


@settings(**SET)
@given(st.integers(0, 2**31 - 1), st.integers(1, 12), st.integers(0, 40), st.integers(1, 4))
def test_argmin_quirk_and_ties(orc, seed, M, N, pool):
    rng = np.random.default_rng(seed)
    q, t = descs(rng, M, pool), descs(rng, max(N, 0), pool)
    mask = (rng.random(N) < 0.3).astype(np.uint8)
    bi, b, s = orc.hamming_argmin(q, t, mask)
    for i in range(M):
        assert (bi[i], b[i], s[i]) == twin.scan_row(q[i], t, mask)
        if bi[i] >= 0:      # ties go to the lowest unmasked column
            d = np.unpackbits(q[i] ^ t, axis=1).sum(1)
            ok = np.where((mask == 0) & (d == b[i]))[0]
            assert bi[i] == ok[0]


@settings(**SET)
@given(st.integers(0, 2**31 - 1), st.integers(1, 14), st.integers(1, 30), st.sampled_from([(15, 0.0), (30, 2.0), (256, 0.0)]))
def test_greedy_claims(orc, seed, M, N, rule):
    rng = np.random.default_rng(seed)
    q, t = descs(rng, M, 3), descs(rng, N, 3)
    assigned = (rng.random(N) < 0.2).astype(np.uint8)
    skip = (rng.random(M) < 0.2).astype(np.uint8)
    bi, b, s, acc, asg = orc.match_greedy(q, t, assigned, rule[0], rule[1], q_skip=skip)
    ref, ref_asg = twin.greedy(q, t, assigned, rule[0], rule[1], skip)
    assert np.array_equal(np.stack([bi, b, s, acc], 1), ref)
    assert np.array_equal(asg, ref_asg)
    won = bi[acc == 1]
    assert len(set(won.tolist())) == len(won)            # a column is claimed at most once
    assert not assigned[won].any()                       # and never a pre-assigned one


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def brute_score(img, x, y, t0=20):
    v = int(img[y, x])
    d = [v - int(img[y + dy, x + dx]) for dx, dy in RING]

    def corner(t):
        return any(all(d[(s + i) % 16] > t for i in range(9)) or all(d[(s + i) % 16] < -t for i in range(9))
                   for s in range(16))
    if not corner(t0):
        return 0
    t = t0
    while corner(t + 1):
        t += 1
    return t


@settings(**SET)
@given(st.integers(0, 2**31 - 1), st.sampled_from(["noise", "steps", "flat"]))
def test_fast_score_closed_form(orc, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        img = rng.integers(0, 256, (9, 9), dtype=np.uint8)
    elif kind == "steps":
        img = np.where(rng.random((9, 9)) < 0.5, 40, 200).astype(np.uint8)
        img[4, 4] = rng.integers(0, 256)
    else:
        img = np.full((9, 9), int(rng.integers(0, 256)), np.uint8)
    img = np.ascontiguousarray(img)
    got = orc.lib().orc_fast_score_at(img.ctypes.data_as(C.c_void_p), 9, 9, 9, 4, 4)
    assert got == brute_score(img, 4, 4)


@settings(max_examples=10, deadline=None)
@given(st.integers(0, 2**31 - 1))
def test_fast_nms_is_strict_local_maximum(orc, seed):
    import util
    img = util.blocky_image(seed % 100000, 100, 90, cells=(3, 7, 19))
    lib = orc.lib()
    score = np.zeros((90, 100), np.int32)
    for y in range(3, 87):
        for x in range(3, 97):
            score[y, x] = lib.orc_fast_score_at(img.ctypes.data_as(C.c_void_p), 100, 90, 100, x, y)
    got = {(int(x), int(y)): int(s) for x, y, s in orc.fast_corners(img, border=31)}
    want = {}
    for y in range(31, 90 - 31):
        for x in range(31, 100 - 31):
            s = score[y, x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if s > 0 and (nb < s).all():
                want[(x, y)] = int(s)
    assert got == want
    keys = list(got)                                      # raster order
    assert keys == sorted(keys, key=lambda p: (p[1], p[0]))


def py_resize(src, dw, dh):
    """cv::resize INTER_LINEAR for 8-bit single channel, fixed point, written from the formula."""
    sh, sw = src.shape

    def tab(d, s):
        scale = s / d
        ofs, a = [], []
        for i in range(d):
            f = np.float32((i + 0.5) * scale - 0.5)
            k = int(np.floor(f)); f = np.float32(f - np.float32(k))
            if k < 0:
                k, f = 0, np.float32(0)
            if k >= s - 1:
                k, f = s - 1, np.float32(0)
            ofs.append(k)
            a.append((int(np.rint(np.float32(1 - f) * np.float32(2048))), int(np.rint(f * np.float32(2048)))))
        return ofs, a
    xo, xa = tab(dw, sw)
    yo, ya = tab(dh, sh)
    out = np.zeros((dh, dw), np.uint8)
    for y in range(dh):
        r0, r1 = src[yo[y]].astype(np.int64), src[min(yo[y] + 1, sh - 1)].astype(np.int64)
        for x in range(dw):
            x0, x1 = xo[x], min(xo[x] + 1, sw - 1)
            S0 = r0[x0] * xa[x][0] + r0[x1] * xa[x][1]
            S1 = r1[x0] * xa[x][0] + r1[x1] * xa[x][1]
            out[y, x] = np.clip((((ya[y][0] * (S0 >> 4)) >> 16) + ((ya[y][1] * (S1 >> 4)) >> 16) + 2) >> 2, 0, 255)
    return out


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 2**31 - 1), st.integers(8, 40), st.integers(8, 30))
def test_resize_fixed_point(orc, seed, sw, sh):
    rng = np.random.default_rng(seed)
    src = np.ascontiguousarray(rng.integers(0, 256, (sh, sw), dtype=np.uint8))
    dw, dh = int(np.rint(np.float32(sw) / np.float32(1.2))), int(np.rint(np.float32(sh) / np.float32(1.2)))
    dst = np.zeros((dh, dw), np.uint8)
    orc.lib().orc_resize_linear_u8(src.ctypes.data_as(C.c_void_p), sw, sh, sw, dst.ctypes.data_as(C.c_void_p), dw, dh, dw)
    assert np.array_equal(dst, py_resize(src, dw, dh))
