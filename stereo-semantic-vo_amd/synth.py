"""synth-kitti: seeded synthetic stereo sequences with exact ground truth.

There is no KITTI data offline, so benchmarks and the
tracking tests use this renderer: a KITTI-00-shaped camera (1241x376, intrinsics of
the reference's Stereo/KITTI00-02.yaml:8-11,25, baseline bf/fx = 0.5372 m) drives
1 m/frame forward with 0.2 deg/frame yaw (the step length of the reference's
Stereo/01.txt ground truth) between a textured ground plane and a textured canopy
plane.  Both views are rendered by exact ray casting of a procedural world texture
(smooth shading + sparse sharp blocks at five scales with distance LOD), so stereo and
temporal geometry are exact; per-pixel sensor noise is hash-seeded.  The texture
statistics are calibrated to the only real street images available here
(Thirdparty/libelas/img/urban*.pgm): ~3-4 % of pixels pass FAST-9 at threshold 20.

Pure torch, device-agnostic (CPU in the tests, GPU in bench.py).  Ground truth poses are
written as KITTI 12-float rows (layout of Stereo/01.txt; main.cpp:141-146 writers).
"""
import math

import torch

KITTI00 = dict(W=1241, H=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448)
BASE_SEED = 0x5EED0000


def _hash01(ix, iz, a, b):
    """uint32 avalanche hash of integer lattice coordinates -> [0,1)."""
    M = 0xFFFFFFFF
    h = (ix * 0x9E3779B1 + iz * 0x85EBCA77 + a * 0xC2B2AE3D + b * 0x27D4EB2F) & M
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & M
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & M
    h = h ^ (h >> 15)
    return (h & 0xFFFFFF).to(torch.float32) / float(1 << 24)


def _texture(px, pz, plane, depth, fx, seed):
    """World texture at (px, pz) on plane id `plane`, seen at distance `depth`."""
    s = seed & 0xFFFF
    # smooth shading: bilinear value noise on a 3 m lattice
    g = 3.0
    fxx, fzz = px / g, pz / g
    ix0, iz0 = torch.floor(fxx), torch.floor(fzz)
    tx, tz = fxx - ix0, fzz - iz0
    ix0, iz0 = ix0.to(torch.int64), iz0.to(torch.int64)
    pl = torch.full_like(ix0, plane)
    sd = torch.full_like(ix0, 1000 + s)
    v00 = _hash01(ix0, iz0, pl, sd); v10 = _hash01(ix0 + 1, iz0, pl, sd)
    v01 = _hash01(ix0, iz0 + 1, pl, sd); v11 = _hash01(ix0 + 1, iz0 + 1, pl, sd)
    base = (v00 * (1 - tx) + v10 * tx) * (1 - tz) + (v01 * (1 - tx) + v11 * tx) * tz
    out = 100.0 + 70.0 * (base - 0.5)
    # sparse sharp blocks at five scales, faded out when a cell covers < ~4 px
    for o, (cell, amp, dens) in enumerate(((0.13, 55.0, 0.10), (0.40, 60.0, 0.14), (1.10, 50.0, 0.18),
                                           (3.10, 45.0, 0.25), (8.70, 45.0, 0.30))):
        ix = torch.floor(px / cell + 0.37 * (o + 1)).to(torch.int64)   # de-align the lattices
        iz = torch.floor(pz / cell + 0.61 * (o + 1)).to(torch.int64)
        oo = torch.full_like(ix, o)
        h1 = _hash01(ix, iz, oo + 8 * plane, torch.full_like(ix, s))
        h2 = _hash01(ix, iz, oo + 8 * plane, torch.full_like(ix, s + 77))
        cell_px = cell * fx / depth
        wgt = torch.clamp((cell_px - 3.0) / 3.0, 0.0, 1.0)
        out = out + torch.where(h1 < dens, (h2 - 0.5) * 2.0 * amp, torch.zeros_like(h2)) * wgt
    return out


def trajectory(n_frames, step=1.0, yaw_deg=0.2, device="cpu"):
    """T_wc (n,4,4) float64: camera-to-world; camera x right, y down, z forward."""
    T = torch.zeros((n_frames, 4, 4), dtype=torch.float64, device=device)
    x = z = 0.0
    for k in range(n_frames):
        a = math.radians(yaw_deg) * k
        c, s = math.cos(a), math.sin(a)
        T[k] = torch.tensor([[c, 0, s, x], [0, 1, 0, 0], [-s, 0, c, z], [0, 0, 0, 1]], dtype=torch.float64)
        x += step * s
        z += step * c
    return T


def kitti_rows(T_wc):
    """KITTI ground-truth layout: 12 floats = first three rows of T_wc."""
    return T_wc[:, :3, :].reshape(-1, 12)


def render_view(T_wc, cam, eye_offset_x, frame, view, seed=BASE_SEED, device="cpu", noise_sigma=2.0):
    W, H = cam["W"], cam["H"]
    dev = torch.device(device)
    T = T_wc.to(dev, torch.float32)
    R, C = T[:3, :3], T[:3, 3] + T[:3, 0] * eye_offset_x
    v, u = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32),
                          torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    dcx, dcy = (u - cam["cx"]) / cam["fx"], (v - cam["cy"]) / cam["fy"]
    dwx = R[0, 0] * dcx + R[0, 1] * dcy + R[0, 2]
    dwy = R[1, 0] * dcx + R[1, 1] * dcy + R[1, 2]
    dwz = R[2, 0] * dcx + R[2, 1] * dcy + R[2, 2]
    y_ground, y_canopy = 1.65, -4.0
    eps = 1e-6
    down = dwy > eps
    up = dwy < -eps
    t = torch.where(down, (y_ground - C[1]) / torch.where(down, dwy, torch.ones_like(dwy)),
                    torch.where(up, (y_canopy - C[1]) / torch.where(up, dwy, -torch.ones_like(dwy)),
                                torch.full_like(dwy, 1e6)))
    t = torch.clamp(t, 0.1, 1e6)
    px, pz = C[0] + t * dwx, C[2] + t * dwz
    img_g = _texture(px, pz, 0, t, cam["fx"], seed)
    img_c = _texture(px, pz, 1, t, cam["fx"], seed)
    img = torch.where(down, img_g, torch.where(up, img_c, torch.full_like(img_g, 128.0)))
    # distance haze keeps the far field (sub-pixel disparity) low-contrast
    haze = torch.clamp(t / 300.0, 0.0, 1.0)
    img = img * (1 - haze) + 128.0 * haze
    if noise_sigma > 0:
        ui, vi = u.to(torch.int64), v.to(torch.int64)
        fr = torch.full_like(ui, frame * 2 + view)
        n = (_hash01(ui, vi, fr, torch.full_like(ui, 11)) + _hash01(ui, vi, fr, torch.full_like(ui, 23)) +
             _hash01(ui, vi, fr, torch.full_like(ui, 37)) + _hash01(ui, vi, fr, torch.full_like(ui, 41)) - 2.0)
        img = img + n * (noise_sigma / math.sqrt(4.0 / 12.0))
    return torch.clamp(torch.round(img), 0, 255).to(torch.uint8)


def render_pair(T_wc, cam=KITTI00, frame=0, seed=BASE_SEED, device="cpu"):
    b = cam["bf"] / cam["fx"]
    L = render_view(T_wc, cam, 0.0, frame, 0, seed, device)
    R = render_view(T_wc, cam, b, frame, 1, seed, device)
    return L, R


def render_sequence(n_frames, cam=KITTI00, seed=BASE_SEED, device="cpu", start=0):
    """(L, R) uint8 tensors (n,H,W) + ground-truth T_wc (n,4,4) for frames start..start+n-1."""
    T = trajectory(start + n_frames, device="cpu")[start:]
    Ls, Rs = [], []
    for k in range(n_frames):
        L, R = render_pair(T[k], cam, start + k, seed, device)
        Ls.append(L); Rs.append(R)
    return torch.stack(Ls), torch.stack(Rs), T
