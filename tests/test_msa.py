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


# ---- the stages that consume the tree: TreeDp, WTA, LRcheck ----------------------------------------------
def random_tree(rng, N, chain=0.0):
    """A random spanning tree as MSA holds it: BFS order from the root and per-node children in a fixed (here
    random) order, with edge weights 0..255.  `chain` > 0 makes long paths (deep trees)."""
    parent = np.full(N, -1, np.int64)
    for v in range(1, N):
        parent[v] = v - 1 if rng.random() < chain else rng.integers(0, v)
    kids = [[] for _ in range(N)]
    for v in rng.permutation(np.arange(1, N)):
        kids[parent[v]].append(int(v))
    child_ptr = np.zeros(N + 1, np.int32); child = []; 
    for u in range(N):
        child_ptr[u + 1] = child_ptr[u] + len(kids[u]); child += kids[u]
    child = np.array(child, np.int32) if child else np.zeros(0, np.int32)
    child_c = rng.integers(0, 256, len(child)).astype(np.uint8)
    seq, q = [], [0]
    while q:
        nq = []
        for u in q:
            seq.append(u); nq += kids[u]
        q = nq
    # BFS by queue (the reference's getSeq) visits level by level with the children in chain order: same thing
    return np.array(seq, np.int32), child_ptr, child, child_c


def test_oracle_tree_dp_on_a_path_is_the_closed_form():
    """On a 3-node path the two passes have a hand-checkable closed form (float32 storage, float64 products)."""
    E = ob.msa_exp_table(0.1)
    cost = np.array([[1.0], [2.0], [4.0]], np.float32)
    seq = np.array([0, 1, 2], np.int32); cp = np.array([0, 1, 2, 2], np.int32); ch = np.array([1, 2], np.int32)
    cc = np.array([10, 20], np.uint8)
    A = ob.msa_tree_dp(cost, seq, cp, ch, cc, 0, E)
    w1, w2 = E[10], E[20]
    up2 = np.float32(4.0); up1 = np.float32(2.0 + w2 * float(up2)); up0 = np.float32(1.0 + w1 * float(up1))
    a0 = up0; a1 = np.float32(w1 * float(a0) + (1 - w1 * w1) * float(up1)); a2 = np.float32(w2 * float(a1) + (1 - w2 * w2) * float(up2))
    assert A[:, 0].tolist() == [a0, a1, a2]


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,chain", [(1, 5, 0.0), (2, 49, 0.0), (5000, 49, 0.0), (20000, 17, 0.9), (3000, 256, 0.5)])
def test_gpu_tree_dp_equals_oracle(pkg, N, D, chain):
    rng = np.random.default_rng(N + D)
    seq, cp, ch, cc = random_tree(rng, N, chain)
    cost = rng.random((N, D), np.float32) * 3
    s = pkg.Svo(640, 240)
    for o in (0.1, 0.05):
        g = s.msa_tree_dp(cost, seq, cp, ch, cc, 0, o)
        r = ob.msa_tree_dp(cost, seq, cp, ch, cc, 0, ob.msa_exp_table(o))
        assert g.tobytes() == r.tobytes()
    s.close()


@pytest.mark.gpu
def test_gpu_tree_dp_level_launch_path_equals_oracle(pkg, monkeypatch):
    """TreeDp runs as one launch per aggregation (a workgroup per disparity walks the levels, k_msa_dp_bfs); the kernels
    with one launch per level stay for trees whose widest level does not fit LDS - forced here."""
    monkeypatch.setenv("SVO_MSA_LEVEL_LAUNCHES", "1")
    rng = np.random.default_rng(77)
    seq, cp, ch, cc = random_tree(rng, 6000, 0.7)
    cost = rng.random((6000, 33), np.float32) * 3
    s = pkg.Svo(640, 240)
    g = s.msa_tree_dp(cost, seq, cp, ch, cc, 0, 0.1)
    assert g.tobytes() == ob.msa_tree_dp(cost, seq, cp, ch, cc, 0, ob.msa_exp_table(0.1)).tobytes()
    monkeypatch.delenv("SVO_MSA_LEVEL_LAUNCHES")
    assert s.msa_tree_dp(cost, seq, cp, ch, cc, 0, 0.1).tobytes() == g.tobytes()
    s.close()


@pytest.mark.gpu
def test_gpu_wta_and_lrcheck_equal_oracle(pkg):
    rng = np.random.default_rng(9)
    H, W, D = 60, 97, 49
    costA = rng.random((H, W, D), np.float32)
    costA[rng.random((H, W)) < 0.3, 7] = 0.0; costA[rng.random((H, W)) < 0.3, 3] = 0.0      # ties: the first minimum wins
    s = pkg.Svo(640, 240)
    assert np.array_equal(s.msa_wta(costA, H, W), ob.msa_wta(costA, H, W))
    d1 = rng.integers(0, D, (H, W)).astype(np.uint8)
    d2 = np.where(rng.random((H, W)) < 0.5, np.roll(d1, -3, 1), rng.integers(0, D, (H, W))).astype(np.uint8)
    d1[:, 10:40] = 3; d2[:, 7:37] = 3
    gc, gm = s.msa_lrcheck(d1, d2, D)
    rc, rm = ob.msa_lrcheck(d1, d2, D)
    assert gm.sum() > 100 and np.array_equal(gm, rm) and gc.tobytes() == rc.tobytes()
    with pytest.raises(pkg.SvoError):
        s.msa_tree_dp(costA.reshape(-1, D)[:4], [1, 0, 2, 3], [0, 3, 3, 3, 3], [1, 2, 3], [0, 0, 0], 0)   # seq[0] != root
    s.close()


# ---- the aggregation tree (build, minimum arborescence, region merging, BFS order) and MSA::solve -----------
def shifted_colour_pair(W, H, shift):
    """Left = right shifted by `shift` columns: the disparity MSA has to recover is `shift` wherever the match exists."""
    L, _ = colour_pair(W + shift, H)
    return np.ascontiguousarray(L[:, :W]), np.ascontiguousarray(L[:, shift:shift + W])


def tree_inputs(bgr):
    o = ob.msa_init(bgr, bgr, 2)
    return o["m3L"], o["r_graL"], o["c_graL"]


@pytest.mark.parametrize("W,H,flat", [(64, 48, False), (200, 120, False), (96, 64, True)])
def test_tree_order_has_the_property_the_one_launch_sweep_relies_on(pkg, W, H, flat):
    """k_msa_dp_bfs addresses the nodes of a tree by their position in level order and expects every node's children at
    consecutive positions of the next level, in child-list order (the host checks this per tree and would fall back to the
    launch-per-level kernels).  The breadth-first `seq` the tree builder returns must give exactly that."""
    if flat:
        bgr = np.full((H, W, 3), 90, np.uint8)
    else:
        bgr, _ = colour_pair(W, H)
    m3, rg, cg = tree_inputs(bgr)
    root, seq, cp, ch, cc = pkg.msa_tree(m3, rg, cg)
    N = W * H
    depth = np.zeros(N, np.int64)
    for u in seq:                                   # parents come before children
        depth[ch[cp[u]:cp[u + 1]]] = depth[u] + 1
    order = seq[np.argsort(depth[seq], kind="stable")]          # level order: by depth, seq order inside a level
    nxt = 1
    for u in order:
        kids = ch[cp[u]:cp[u + 1]]
        assert np.array_equal(order[nxt:nxt + len(kids)], kids)
        nxt += len(kids)
    assert nxt == N and order[0] == root


@pytest.mark.parametrize("W,H", [(5, 5), (64, 48), (200, 120)])
def test_product_tree_equals_literal_restatement(pkg, W, H):
    """svo_msa_tree (host code of the product, array heaps and union-find written for speed) against orc_msa_tree,
    the pointer-for-pointer restatement of MSA.cpp:141-927: same root, same BFS order, same child lists and weights."""
    bgr, _ = colour_pair(W, H)
    m3, rg, cg = tree_inputs(bgr)
    a = pkg.msa_tree(m3, rg, cg)
    b = ob.msa_tree(m3, rg, cg)
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)
    root, seq, cp, ch, cc = a
    N = W * H
    assert seq[0] == root and sorted(seq.tolist()) == list(range(N)) and cp[-1] == N - 1        # a spanning tree
    assert sorted(ch.tolist()) == [v for v in range(N) if v != root]


def test_product_tree_on_a_flat_image(pkg):
    """All weights equal: every tie-break of the heaps and of the unstable sort is on the path."""
    m3 = np.full((40, 56, 3), 90, np.uint8)
    z = np.zeros((40, 56))
    a = pkg.msa_tree(m3, z, z); b = ob.msa_tree(m3, z, z)
    assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))


def test_oracle_solve_recovers_a_known_shift():
    """Parity unpinned (the reference's MSA needs OpenCV to build): the restatement is at least checked for what the
    algorithm must do, recover a constant shift away from the left border."""
    L, R = shifted_colour_pair(120, 64, 7)
    d = ob.msa_solve(L, R, 16, 1)
    assert (d[:, 24:] == 7).mean() > 0.97
    assert np.array_equal(ob.msa_solve(L, R, 16, 3), (d * 3).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,d,shift", [(120, 64, 16, 7), (200, 120, 48, 11), (321, 97, 31, 0), (1241, 376, 48, 0)])
def test_gpu_solve_equals_oracle(pkg, W, H, d, shift):
    """svo_msa_solve (GPU stages + host trees, volumes resident in HBM) is bit-identical to orc_msa_solve."""
    if W > 1000:      # the full KITTI frame (the urban fixture is exactly that big): trees ~1000 levels deep
        gl, gr = util.urban_pair(W, H, 0, 0)
        L = np.ascontiguousarray(np.repeat(gl[:, :, None], 3, 2)); R = np.ascontiguousarray(np.repeat(gr[:, :, None], 3, 2))
    else:
        L, R = shifted_colour_pair(W, H, shift) if shift else colour_pair(W, H)
    s = pkg.Svo(640, 240)
    g = s.msa_solve(L, R, d, 1)
    r = ob.msa_solve(L, R, d, 1)
    assert np.array_equal(g, r), int((g != r).sum())
    if shift:
        assert (g[:, 3 * shift:] == shift).mean() > 0.95
    assert np.array_equal(s.msa_solve(L, R, d, 2), (r * 2).astype(np.uint8))
    with pytest.raises(pkg.SvoError):
        s.msa_solve(L[:4], R[:4], d, 1)
    with pytest.raises(pkg.SvoError):
        s.msa_solve(L, R, 256, 1)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [5, 20])       # one chunk; two chunks on two lanes (16 + 4 frames)
def test_gpu_batch_equals_single_solves(pkg, B):
    """svo_msa_batch_dev (frames solved together: shared level sweeps, stacked per-pixel stages) == svo_msa_solve per
    frame on the B = G = R colour images, for frames of different content (different tree depths in one chunk)."""
    import torch
    W, H, d = 200, 120, 32
    dev = torch.device("cuda", 0)
    pitch = 256
    Ls, Rs = [], []
    for b in range(B):
        L, R = util.urban_pair(W, H, 100 + 45 * b, 40 + 10 * b)
        Ls.append(L); Rs.append(R)
    Ls[3] = np.full((H, W), 80, np.uint8); Rs[3] = np.full((H, W), 80, np.uint8)     # a flat frame: a very different tree
    dL = torch.zeros((B, H, pitch), dtype=torch.uint8, device=dev); dR = torch.zeros_like(dL)
    dL[:, :, :W] = torch.from_numpy(np.stack(Ls)).to(dev); dR[:, :, :W] = torch.from_numpy(np.stack(Rs)).to(dev)
    out = torch.zeros((B, H, W), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    s = pkg.Svo(640, 240)
    s.msa_batch_dev(dL.data_ptr(), dR.data_ptr(), pitch, W, H, B, out.data_ptr(), d)
    got = out.cpu().numpy()
    g2c = lambda g: np.ascontiguousarray(np.repeat(g[:, :, None], 3, 2))
    for b in range(B):
        want = s.msa_solve(g2c(Ls[b]), g2c(Rs[b]), d, 1).astype(np.float32)
        assert np.array_equal(got[b], want), b
    with pytest.raises(pkg.SvoError):
        s.msa_batch_dev(dL.data_ptr(), dR.data_ptr(), W - 1, W, H, B, out.data_ptr(), d)
    s.close()
