"""ctypes binding of the CPU oracle (oracle/libsvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg - never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class LmStats(C.Structure):
    _fields_ = [("n_edges", C.c_int32), ("iterations", C.c_int32), ("trials_total", C.c_int32),
                ("terminated", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double)]


class PnpStats(C.Structure):  # orc_pnp_stats
    _fields_ = [("n_points", C.c_int32), ("n_inliers", C.c_int32), ("best_hypothesis", C.c_int32),
                ("ok", C.c_int32), ("iterations", C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "libsvo_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_pyramid_size.restype = C.c_int64
        _LIB.orc_level_offset.restype = C.c_int64
        _LIB.orc_harris_at.restype = C.c_int64
        _LIB.orc_harris_to_float.restype = C.c_float
        _LIB.orc_harris_to_float.argtypes = [C.c_int64]
        _LIB.orc_fast_atan2.restype = C.c_float
        _LIB.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _LIB.orc_ic_angle.restype = C.c_float
        _LIB.orc_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def geometry(W, H, nfeatures=500):
    w = np.zeros(8, np.int32); h = np.zeros(8, np.int32)
    s = np.zeros(8, np.float32); q = np.zeros(8, np.int32)
    lib().orc_geometry(W, H, nfeatures, _p(w), _p(h), _p(s), _p(q))
    return w, h, s, q


def umax():
    u = np.zeros(16, np.int32)
    lib().orc_umax(_p(u))
    return u


def build_pyramid(gray):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    pyr = np.zeros(lib().orc_pyramid_size(W, H), np.uint8)
    lib().orc_build_pyramid(_p(gray), W, H, W, _p(pyr))
    return pyr


def pyramid_levels(pyr, W, H):
    w, h, _, _ = geometry(W, H)
    out, off = [], 0
    for l in range(8):
        out.append(pyr[off:off + w[l] * h[l]].reshape(h[l], w[l]))
        off += int(w[l]) * int(h[l])
    return out


def fast_corners(img, border=31):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = (w // 2 + 1) * (h // 2 + 1)
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast_corners(_p(img), w, h, w, 20, border, _p(out), cap)
    return out[:n].copy()


def orb_extract(gray, nfeatures=500, want_pyramid=False):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    kp = np.zeros(nfeatures, KP_DTYPE)
    desc = np.zeros((nfeatures, 32), np.uint8)
    pyr = np.zeros(lib().orc_pyramid_size(W, H), np.uint8) if want_pyramid else None
    n = lib().orc_orb_extract(_p(gray), W, H, W, nfeatures, _p(kp), _p(desc), _p(pyr))
    if want_pyramid:
        return kp[:n].copy(), desc[:n].copy(), pyr
    return kp[:n].copy(), desc[:n].copy()


def stereo_frame(grayL, grayR, bf, fx, nfeatures=500):
    grayL = np.ascontiguousarray(grayL, np.uint8); grayR = np.ascontiguousarray(grayR, np.uint8)
    H, W = grayL.shape
    kpL = np.zeros(nfeatures, KP_DTYPE); dL = np.zeros((nfeatures, 32), np.uint8)
    kpR = np.zeros(nfeatures, KP_DTYPE); dR = np.zeros((nfeatures, 32), np.uint8)
    uR = np.zeros(nfeatures, np.float32); depth = np.zeros(nfeatures, np.float32)
    nL = C.c_int32(0); nR = C.c_int32(0)
    lib().orc_stereo_frame(_p(grayL), W, _p(grayR), W, W, H, nfeatures, C.c_float(bf), C.c_float(fx),
                           _p(kpL), _p(dL), C.byref(nL), _p(uR), _p(depth), _p(kpR), _p(dR),
                           C.byref(nR))
    nl, nr = nL.value, nR.value
    return dict(kpL=kpL[:nl].copy(), dL=dL[:nl].copy(), uR=uR[:nl].copy(), depth=depth[:nl].copy(),
                kpR=kpR[:nr].copy(), dR=dR[:nr].copy())


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    return np.array([lib().orc_descriptor_distance(_p(a[i]), _p(b[i])) for i in range(len(a))], np.int32)


def hamming_argmin(q, t, t_mask=None):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    M, N = len(q), len(t)
    bi = np.zeros(M, np.int32); b = np.zeros(M, np.int32); s = np.zeros(M, np.int32)
    m = None if t_mask is None else np.ascontiguousarray(t_mask, np.uint8)
    lib().orc_hamming_argmin(_p(q), M, _p(t), N, _p(m), _p(bi), _p(b), _p(s))
    return bi, b, s


def match_greedy(q, t, assigned, max_dist, ratio, q_skip=None):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    M, N = len(q), len(t)
    assigned = np.ascontiguousarray(assigned, np.uint8).copy()
    sk = None if q_skip is None else np.ascontiguousarray(q_skip, np.uint8)
    bi = np.zeros(M, np.int32); b = np.zeros(M, np.int32); s = np.zeros(M, np.int32)
    acc = np.zeros(M, np.uint8)
    lib().orc_match_greedy(_p(q), _p(sk), M, _p(t), N, _p(assigned), max_dist, C.c_float(ratio),
                           _p(bi), _p(b), _p(s), _p(acc))
    return bi, b, s, acc, assigned


def bf_match(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    M, N = len(q), len(t)
    ti = np.zeros(M, np.int32); d = np.zeros(M, np.int32); keep = np.zeros(M, np.uint8)
    lib().orc_bf_match(_p(q), M, _p(t), N, _p(ti), _p(d), _p(keep))
    return ti, d, keep


def huber(e, delta):
    rho = np.zeros(3)
    lib().orc_huber(C.c_double(e), C.c_double(delta), _p(rho))
    return rho


def exp_coeffs(t, use_libm=False):
    abc = np.zeros(3)
    lib().orc_exp_coeffs(C.c_double(t), _p(abc), 1 if use_libm else 0)
    return abc


def se3_exp_matrix(upd):
    upd = np.ascontiguousarray(upd, np.float64); T = np.zeros(16)
    lib().orc_se3_exp_matrix(_p(upd), _p(T))
    return T.reshape(4, 4)


def se3_update(upd, T):
    upd = np.ascontiguousarray(upd, np.float64); T = np.ascontiguousarray(T, np.float64).reshape(16).copy()
    lib().orc_se3_update(_p(upd), _p(T))
    return T.reshape(4, 4)


def pose_opt(Xw, obs, K, T, trace_cap=128):
    Xw = np.ascontiguousarray(Xw, np.float64); obs = np.ascontiguousarray(obs, np.float64)
    K = np.ascontiguousarray(K, np.float64); T = np.ascontiguousarray(T, np.float64).reshape(16).copy()
    st = LmStats(); trace = np.zeros((trace_cap, 8))
    nt = lib().orc_pose_opt(_p(Xw), _p(obs), len(Xw), _p(K), _p(T), C.byref(st), _p(trace), trace_cap)
    return T.reshape(4, 4), st, trace[:nt].copy()


def pnp_ransac(Xw, obs, K, T_fallback, rng_state=0, refine=0):
    """cv::solvePnPRansac(..., false, 100, 8.0, 0.99) restated; rng_state 0 = OpenCV's (uint64)-1."""
    Xw = np.ascontiguousarray(Xw, np.float64); obs = np.ascontiguousarray(obs, np.float64)
    K = np.ascontiguousarray(K, np.float64); Tp = np.ascontiguousarray(T_fallback, np.float64).reshape(16)
    T = np.zeros(16); mask = np.zeros(max(len(Xw), 1), np.uint8); st = PnpStats()
    lib().orc_solvepnp_ransac(_p(Xw), _p(obs), len(Xw), _p(K), _p(Tp), C.c_uint64(rng_state), int(refine), _p(T), _p(mask),
                              C.byref(st))
    return T.reshape(4, 4), mask[:len(Xw)], st


def epnp5(Xw5, uv5, K):
    Xw5 = np.ascontiguousarray(Xw5, np.float64).reshape(15); uv5 = np.ascontiguousarray(uv5, np.float64).reshape(10)
    K = np.ascontiguousarray(K, np.float64); R = np.zeros(9); t = np.zeros(3)
    lib().orc_epnp5(_p(Xw5), _p(uv5), _p(K), _p(R), _p(t))
    return R.reshape(3, 3), t


def disp2depth(disp, bf):
    disp = np.ascontiguousarray(disp, np.float32); out = np.zeros_like(disp)
    lib().orc_disp2depth(_p(disp), disp.size, C.c_float(bf), _p(out))
    return out


def unproject(uvz, cam, Rwc, twc):
    uvz = np.ascontiguousarray(uvz, np.float32).reshape(-1, 3)
    Rwc = np.ascontiguousarray(Rwc, np.float32).reshape(9); twc = np.ascontiguousarray(twc, np.float32).reshape(3)
    out = np.zeros_like(uvz)
    fx, fy, cx, cy = (C.c_float(float(v)) for v in cam[:4])
    lib().orc_unproject(_p(uvz), len(uvz), fx, fy, cx, cy, _p(Rwc), _p(twc), _p(out))
    return out


TRACK_DTYPE = np.dtype([("Tcw", "<f4", (16,)), ("frame_id", "<i4"), ("n_kp", "<i4"),
                        ("n_stereo", "<i4"), ("n_match_pass1", "<i4"), ("n_match_pass2", "<i4"),
                        ("n_pnp_inliers", "<i4"), ("n_lm_edges", "<i4"), ("n_new_mappoints", "<i4"),
                        ("n_local_map", "<i4"), ("lm_iterations", "<i4"), ("reserved", "<i4", (2,))])


# orc_tail_debug (svo_oracle.h); 4 bytes of padding in front of the doubles
TAIL_DEBUG_DTYPE = np.dtype([("match_gid", "<i4", (512,)), ("new_gid", "<i4", (512,)), ("pnp", "<i4", (5,)), ("_pad", "<i4"),
                             ("T_pnp", "<f8", (16,))])


class Tracker:
    """orc_track.c: the reference's Tracking::Track loop on the CPU."""

    def __init__(self, W, H, cam, nfeatures=500):
        l = lib()
        l.orc_track_create.restype = C.c_void_p
        l.orc_track_create.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_float] * 5
        l.orc_track_destroy.argtypes = [C.c_void_p]
        l.orc_track_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p]
        self.W, self.H, self.nf = W, H, nfeatures
        self.h = l.orc_track_create(W, H, nfeatures, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"])

    def track(self, grayL, grayR, boxes=None, dense=None, Tcw_force=None):
        """dense: optional H x W float32 disparity map used as the depth source (frame::MB data flow).
        Tcw_force: teacher forcing - the frame reports its own pose, the tracker continues from this one."""
        grayL = np.ascontiguousarray(grayL, np.uint8); grayR = np.ascontiguousarray(grayR, np.uint8)
        res = np.zeros(1, TRACK_DTYPE); cur = np.zeros(self.nf, np.int32)
        bx = None if boxes is None or len(boxes) == 0 else np.ascontiguousarray(boxes, np.int32)
        dn = None if dense is None else np.ascontiguousarray(dense, np.float32)
        self.F = np.zeros(9)
        l = lib()
        l.orc_track_frame_dense.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        if Tcw_force is not None:
            tf = np.ascontiguousarray(Tcw_force, np.float32).reshape(16)
            l.orc_track_force_pose.argtypes = [C.c_void_p, C.c_void_p]
            l.orc_track_force_pose(self.h, _p(tf))
        l.orc_track_frame_dense(self.h, _p(grayL), self.W, _p(grayR), self.W, _p(dn), _p(bx),
                                0 if bx is None else len(bx), _p(res), _p(cur), _p(self.F))
        self.vetoes = l.orc_track_last_vetoes(C.c_void_p(self.h))
        return res[0], cur

    def track_tail(self, kp, desc, depth, boxes=None, Tcw_force=None):
        """orc_track_tail: the ordered tail for given front-end results (kp: KP_DTYPE records, desc: n x 32, depth: n
        floats).  Returns (record, cur_mp, pnp stats dict, T_pnp 4x4); self.match_gid / self.new_gid: per-keypoint map-point
        identities (creation sequence numbers) of the frame."""
        n = len(kp)
        kp = np.ascontiguousarray(kp); desc = np.ascontiguousarray(desc, np.uint8); depth = np.ascontiguousarray(depth, np.float32)
        assert kp.dtype.itemsize == 28 and desc.shape == (n, 32) and depth.shape == (n,)
        res = np.zeros(1, TRACK_DTYPE); cur = np.full(self.nf, -1, np.int32)
        bx = None if boxes is None or len(boxes) == 0 else np.ascontiguousarray(boxes, np.int32)
        tf = None if Tcw_force is None else np.ascontiguousarray(Tcw_force, np.float32).reshape(16)
        self.F = np.zeros(9)
        dbg = np.zeros(1, TAIL_DEBUG_DTYPE)
        l = lib()
        l.orc_track_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_track_tail(self.h, _p(kp), _p(desc), n, _p(depth), _p(bx), 0 if bx is None else len(bx), _p(tf), _p(res),
                         _p(cur), _p(self.F), _p(dbg))
        self.vetoes = l.orc_track_last_vetoes(C.c_void_p(self.h))
        d = dbg[0]
        self.match_gid, self.new_gid = d["match_gid"].copy(), d["new_gid"].copy()
        pnp = dict(n_points=int(d["pnp"][0]), n_inliers=int(d["pnp"][1]), best_hypothesis=int(d["pnp"][2]), ok=int(d["pnp"][3]),
                   iterations=int(d["pnp"][4]))
        return res[0], cur, pnp, d["T_pnp"].reshape(4, 4).copy()

    def close(self):
        if self.h:
            lib().orc_track_destroy(self.h)
            self.h = None


def fundamental_8point(pts1, pts2):
    pts1 = np.ascontiguousarray(pts1, np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(pts2, np.float64).reshape(-1, 2)
    F = np.zeros(9)
    lib().orc_fundamental_8point(_p(pts1), _p(pts2), len(pts1), _p(F))
    return F.reshape(3, 3)


def epipolar_distance(F, last_xy, cur_xy):
    l = lib()
    l.orc_epipolar_distance.restype = C.c_double
    l.orc_epipolar_distance.argtypes = [C.c_void_p] + [C.c_float] * 4
    F = np.ascontiguousarray(F, np.float64).reshape(9)
    return l.orc_epipolar_distance(_p(F), last_xy[0], last_xy[1], cur_xy[0], cur_xy[1])


# --- compiled REFERENCE pieces (oracle/Makefile.ref -> oracle/_ref/) --------------------------------
class ElasParams(C.Structure):
    """Elas::parameters (Thirdparty/libelas/src/elas.h:60-83), bools as int32."""
    _fields_ = [("disp_min", C.c_int32), ("disp_max", C.c_int32), ("support_threshold", C.c_float),
                ("support_texture", C.c_int32), ("candidate_stepsize", C.c_int32),
                ("incon_window_size", C.c_int32), ("incon_threshold", C.c_int32),
                ("incon_min_support", C.c_int32), ("add_corners", C.c_int32), ("grid_size", C.c_int32),
                ("beta", C.c_float), ("gamma", C.c_float), ("sigma", C.c_float), ("sradius", C.c_float),
                ("match_texture", C.c_int32), ("lr_threshold", C.c_int32),
                ("speckle_sim_threshold", C.c_float), ("speckle_size", C.c_int32),
                ("ipol_gap_width", C.c_int32), ("filter_median", C.c_int32),
                ("filter_adaptive_mean", C.c_int32), ("postprocess_only_left", C.c_int32),
                ("subsampling", C.c_int32)]


class _ElasTaps(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("desc2", C.c_void_p), ("support", C.c_void_p),
                ("n_support", C.c_int32), ("cap_support", C.c_int32),
                ("tri1", C.c_void_p), ("tri2", C.c_void_p), ("planes1", C.c_void_p), ("planes2", C.c_void_p),
                ("n_tri1", C.c_int32), ("n_tri2", C.c_int32), ("cap_tri", C.c_int32),
                ("grid1", C.c_void_p), ("grid2", C.c_void_p),
                ("D1_raw", C.c_void_p), ("D2_raw", C.c_void_p), ("D1_lr", C.c_void_p), ("D2_lr", C.c_void_p),
                ("D1_seg", C.c_void_p), ("D2_seg", C.c_void_p), ("D1_gap", C.c_void_p), ("D2_gap", C.c_void_p),
                ("D1_mean", C.c_void_p), ("D2_mean", C.c_void_p),
                ("tri1_in", C.c_void_p), ("tri2_in", C.c_void_p),
                ("n_tri1_in", C.c_int32), ("n_tri2_in", C.c_int32)]


_REF_ELAS = None


def ref_elas_lib():
    """The reference's real libelas (Thirdparty/libelas), built by oracle/Makefile.ref.
    Returns None when oracle/_ref/ has not been built (no /root/reference at build time)."""
    global _REF_ELAS
    so = os.path.join(_HERE, "_ref", "libref_elas.so")
    if _REF_ELAS is None and os.path.exists(so):
        _REF_ELAS = C.CDLL(so)
    return _REF_ELAS


def ref_elas_params(middlebury=False):
    p = ElasParams()
    ref_elas_lib().ref_elas_default_params(int(middlebury), C.byref(p))
    return p


def ref_elas(grayL, grayR, params=None):
    """Elas::process untouched: dense left/right disparity maps (float32, negative = invalid)."""
    L = ref_elas_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_elas.so not built (make -C oracle -f Makefile.ref)")
    params = params or ref_elas_params()
    gl = np.ascontiguousarray(grayL, np.uint8); gr = np.ascontiguousarray(grayR, np.uint8)
    H, W = gl.shape
    Hd, Wd = (H // 2, W // 2) if params.subsampling else (H, W)
    D1 = np.zeros((Hd, Wd), np.float32); D2 = np.zeros((Hd, Wd), np.float32)
    L.ref_elas_process(_p(gl), _p(gr), W, H, W, C.byref(params), _p(D1), _p(D2))
    return D1, D2


def ref_elas_staged(grayL, grayR, params=None, tri1=None, tri2=None, cap_support=60000, cap_tri=120000):
    """The reference's stages run one by one with every intermediate tapped; tri1/tri2 (n,3 int32)
    override the triangle lists of the left/right image (see oracle/ref_elas_wrap.cpp)."""
    L = ref_elas_lib()
    params = params or ref_elas_params()
    gl = np.ascontiguousarray(grayL, np.uint8); gr = np.ascontiguousarray(grayR, np.uint8)
    H, W = gl.shape
    gw = -(-W // params.grid_size); gh = -(-H // params.grid_size)
    o = dict(desc1=np.zeros((H, W, 16), np.uint8), desc2=np.zeros((H, W, 16), np.uint8),
             support=np.zeros((cap_support, 3), np.int32),
             tri1=np.zeros((cap_tri, 3), np.int32), tri2=np.zeros((cap_tri, 3), np.int32),
             planes1=np.zeros((cap_tri, 6), np.float32), planes2=np.zeros((cap_tri, 6), np.float32),
             grid1=np.zeros((gh, gw, params.disp_max + 2), np.int32),
             grid2=np.zeros((gh, gw, params.disp_max + 2), np.int32))
    Hd, Wd = (H // 2, W // 2) if params.subsampling else (H, W)
    for k in ("raw", "lr", "seg", "gap", "mean"):
        o["D1_" + k] = np.zeros((Hd, Wd), np.float32); o["D2_" + k] = np.zeros((Hd, Wd), np.float32)
    t = _ElasTaps()
    for k, a in o.items():
        setattr(t, k, a.ctypes.data)
    t.cap_support = cap_support; t.cap_tri = cap_tri
    keep = []
    for name, tri in (("tri1_in", tri1), ("tri2_in", tri2)):
        if tri is not None:
            a = np.ascontiguousarray(tri, np.int32); keep.append(a)
            setattr(t, name, a.ctypes.data); setattr(t, "n_" + name, len(a))
    D1 = np.zeros((Hd, Wd), np.float32); D2 = np.zeros((Hd, Wd), np.float32)
    rc = L.ref_elas_staged(_p(gl), _p(gr), W, H, W, C.byref(params), C.byref(t), _p(D1), _p(D2))
    assert t.n_support <= cap_support and t.n_tri1 <= cap_tri and t.n_tri2 <= cap_tri
    o["support"] = o["support"][:t.n_support].copy()
    for s in ("1", "2"):
        n = getattr(t, "n_tri" + s)
        o["tri" + s] = o["tri" + s][:n].copy(); o["planes" + s] = o["planes" + s][:n].copy()
    o["D1"] = D1; o["D2"] = D2; o["rc"] = rc
    return o


def ref_elas_delaunay(xy):
    """Triangle ("zQB") as libelas calls it, on n (x, y) integer points; (m, 3) int32 triangles."""
    xy = np.ascontiguousarray(xy, np.int32)
    cap = 4 * len(xy) + 16
    tri = np.zeros((cap, 3), np.int32)
    n = ref_elas_lib().ref_elas_delaunay(_p(xy), len(xy), _p(tri), cap)
    return tri[:n].copy()


def canonical_triangles(tri):
    """Smallest index first (orientation kept), rows sorted - the order svo_elas_delaunay emits."""
    t = np.asarray(tri, np.int32).reshape(-1, 3)
    if len(t) == 0:
        return t
    k = np.argmin(t, 1)
    t = np.stack([np.roll(r, -kk) for r, kk in zip(t, k)])
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def ref_ctmf(img, r):
    """The reference's ctmf() (Thirdparty/MB/ctmf.c, compiled by oracle/Makefile.ref) on an H x W or
    H x W x C uint8 image, with the memsize MSA passes (the whole image, MSA.cpp:58)."""
    so = os.path.join(_HERE, "_ref", "libref_ctmf.so")
    L = C.CDLL(so)
    a = np.ascontiguousarray(img, np.uint8)
    cn = 1 if a.ndim == 2 else a.shape[2]
    H, W = a.shape[:2]
    out = np.zeros_like(a)
    L.ctmf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulong]
    L.ctmf(_p(a), _p(out), W, H, W * cn, W * cn, int(r), cn, max(W * H * cn, 1 << 16))
    return out


def msa_init(bgrL, bgrR, disp=49):
    """orc_msa_init: MSA::init on two H x W x 3 uint8 images -> dict of cost volumes, median images, gradients."""
    a = np.ascontiguousarray(bgrL, np.uint8); b = np.ascontiguousarray(bgrR, np.uint8)
    H, W = a.shape[:2]
    o = dict(costL=np.zeros((H, W, disp), np.float32), costR=np.zeros((H, W, disp), np.float32),
             m3L=np.zeros_like(a), m3R=np.zeros_like(a),
             r_graL=np.zeros((H, W)), c_graL=np.zeros((H, W)), r_graR=np.zeros((H, W)), c_graR=np.zeros((H, W)))
    rc = lib().orc_msa_init(_p(a), _p(b), H, W, int(disp), _p(o["costL"]), _p(o["costR"]), _p(o["m3L"]), _p(o["m3R"]),
                            _p(o["r_graL"]), _p(o["c_graL"]), _p(o["r_graR"]), _p(o["c_graR"]))
    assert rc == 0
    return o


def msa_exp_table(o):
    E = np.zeros(256)
    lib().orc_msa_exp_table.argtypes = [C.c_double, C.c_void_p]
    lib().orc_msa_exp_table(float(o), _p(E))
    return E


def msa_tree_dp(cost, seq, child_ptr, child, child_c, root, Exp):
    cost = np.ascontiguousarray(cost, np.float32)
    N, D = cost.shape
    up = np.zeros_like(cost); A = np.zeros_like(cost)
    lib().orc_msa_tree_dp(_p(cost), N, D, _p(np.ascontiguousarray(seq, np.int32)),
                          _p(np.ascontiguousarray(child_ptr, np.int32)), _p(np.ascontiguousarray(child, np.int32)),
                          _p(np.ascontiguousarray(child_c, np.uint8)), int(root), _p(np.ascontiguousarray(Exp, np.float64)),
                          _p(up), _p(A))
    return A


def msa_wta(costA, n, m):
    costA = np.ascontiguousarray(costA, np.float32)
    D = costA.shape[-1]
    out = np.zeros((n, m), np.uint8)
    lib().orc_msa_wta(_p(costA), n, m, D, _p(out))
    return out


def msa_lrcheck(d1, d2, D):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    n, m = d1.shape
    cost = np.zeros((n, m, D), np.float32); mask = np.zeros((n, m), np.uint8)
    lib().orc_msa_lrcheck(_p(d1), _p(d2), n, m, D, _p(cost), _p(mask))
    return cost, mask


def msa_tree(m_img3, r_gra, c_gra):
    """orc_msa_tree: (root, seq, child_ptr, child, child_c) of the spanning tree MSA aggregates over for one image."""
    a = np.ascontiguousarray(m_img3, np.uint8)
    n, m = a.shape[:2]
    N = n * m
    seq = np.zeros(N, np.int32); cp = np.zeros(N + 1, np.int32); ch = np.zeros(N, np.int32); cc = np.zeros(N, np.uint8)
    root = lib().orc_msa_tree(_p(a), _p(np.ascontiguousarray(r_gra, np.float64)), _p(np.ascontiguousarray(c_gra, np.float64)),
                              n, m, _p(seq), _p(cp), _p(ch), _p(cc))
    assert root >= 0, root
    return root, seq, cp, ch[:N - 1].copy(), cc[:N - 1].copy()


def msa_solve(bgrL, bgrR, d=48, scale=1):
    """orc_msa_solve: MSA::solve(l, r, d, scale) -> H x W uint8 disparity image."""
    a = np.ascontiguousarray(bgrL, np.uint8); b = np.ascontiguousarray(bgrR, np.uint8)
    n, m = a.shape[:2]
    out = np.zeros((n, m), np.uint8)
    rc = lib().orc_msa_solve(_p(a), _p(b), n, m, int(d), int(scale), _p(out))
    assert rc == 0, rc
    return out
