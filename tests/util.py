"""Shared test inputs: real street images (libelas urban1 pair - data files the reference
ships under Thirdparty/libelas/img/) and seeded synthetic images / descriptors."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KITTI_W, KITTI_H = 1241, 376


def read_pgm(path):
    """Binary PGM (P5, maxval 255); header tokens may be split over lines in any way, '#' comments."""
    data = open(path, "rb").read()
    tok, pos = [], 0
    while len(tok) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tok.append(data[pos:end]); pos = end
    assert tok[0] == b"P5" and int(tok[3]) == 255
    w, h = int(tok[1]), int(tok[2])
    return np.frombuffer(data, np.uint8, w * h, pos + 1).reshape(h, w).copy()


def urban_pair(W=KITTI_W, H=KITTI_H, x0=50, y0=8):
    """KITTI-sized crop (same crop in both images keeps the epipolar geometry)."""
    L = read_pgm(os.path.join(GOLDEN, "urban1_left.pgm"))
    R = read_pgm(os.path.join(GOLDEN, "urban1_right.pgm"))
    return (np.ascontiguousarray(L[y0:y0 + H, x0:x0 + W]),
            np.ascontiguousarray(R[y0:y0 + H, x0:x0 + W]))


def blocky_image(seed, W, H, cells=(4, 9, 23, 61)):
    """Multi-octave blocky value noise: sharp edges and corners at several scales."""
    rng = np.random.default_rng(seed)
    img = np.full((H, W), 128.0)
    amp = 70.0
    for c in cells:
        g = rng.uniform(-1, 1, size=(H // c + 2, W // c + 2))
        img += amp * np.kron(g, np.ones((c, c)))[:H, :W]
        amp *= 0.6
    img += rng.normal(0, 2.0, size=(H, W))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def shifted_pair(seed, W, H, disparity=12):
    """Left/right pair with a constant integer disparity (right = left shifted left)."""
    assert 0 <= disparity <= 64
    big = blocky_image(seed, W + 128, H)
    L = big[:, 64:64 + W]
    R = big[:, 64 + disparity:64 + disparity + W]
    return np.ascontiguousarray(L), np.ascontiguousarray(R)


def random_descriptors(seed, n):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def flip_bits(desc, nbits, rng):
    d = desc.copy()
    pos = rng.choice(256, size=nbits, replace=False)
    for p in pos:
        d[p // 8] ^= np.uint8(1 << (p % 8))
    return d


def planted_descriptors(seed, M, N):
    """Train set of N random descriptors and M queries with planted neighbours at the
    distances the reference thresholds on ({0,5,14,15,29,30}), plus ties."""
    rng = np.random.default_rng(seed)
    t = random_descriptors(seed + 1, N)
    q = random_descriptors(seed + 2, M)
    dists = [0, 5, 14, 15, 29, 30]
    for i in range(M):
        j = int(rng.integers(0, N))
        q[i] = flip_bits(t[j], dists[i % len(dists)], rng)
        if i % 7 == 0 and N > 3:          # plant an exact tie at a second column
            j2 = int(rng.integers(0, N))
            t[j2] = t[j]
    return q, t


def pose_problem(seed, n=500, outlier_frac=0.2, sigma=0.5, cam=(718.856, 718.856, 607.1928, 185.2157)):
    """SURVEY 8d pose-opt workload: Z~U[5,80] m, 20% outliers of 50 px, sigma 0.5 px."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = cam
    u = rng.uniform(40, 1200, n); v = rng.uniform(40, 340, n); z = rng.uniform(5, 80, n)
    Xc = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1)
    # true pose: small rotation + ~1 m forward
    ang = np.array([0.004, -0.012, 0.002])
    th = np.linalg.norm(ang); k = ang / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.05, -0.02, -1.0])
    T_true = np.eye(4); T_true[:3, :3] = R; T_true[:3, 3] = t
    Xw = (R.T @ (Xc - t).T).T
    obs = np.stack([u, v], 1) + rng.normal(0, sigma, (n, 2))
    out = rng.random(n) < outlier_frac
    obs[out] += rng.choice([-1, 1], (out.sum(), 2)) * 50.0
    Xw = Xw.astype(np.float32).astype(np.float64)   # world points are CV_32F in the reference
    obs = obs.astype(np.float32).astype(np.float64)
    return Xw, obs, np.array(cam, np.float64), T_true
