"""stereo-semantic-vo_amd - MI355X-native stereo-VO tracking front end (host binding).

Thin ctypes layer over the C-ABI shared library ``libsvo_hip.so`` (include/svo.h).
There is NO CPU fallback: if the library is missing, or no HIP device is usable,
construction fails loudly.  The directory name carries a hyphen (it mirrors the
reference repo's name), so import it through ``svo_loader.load()`` at the repo root
or ``importlib`` - the module registers itself as ``stereo_semantic_vo_amd``.

Class surface mirrors the reference's hot-path seams:
  Svo.orb_extract       <- frame::featuredetect            (src/frame.cc:75-79)
  Svo.stereo_frame      <- frame::MB + computekeypoint_r + disp2Depth
                                                            (src/Tracking.cc:226-228)
  Svo.descriptor_distance / hamming_argmin / match_greedy / bf_match
                        <- pnpmatch::DescriptorDistance, poseEstimationPnP passes,
                           find_feature_matches             (src/pnpmatch.cc)
  Svo.pnp_ransac        <- cv::solvePnPRansac call          (src/pnpmatch.cc:227)
  Svo.pose_opt          <- Optimizer::PoseOptimization      (src/Optimizer.cc:15-86)
  Svo.track_*           <- Tracking::Track                  (src/Tracking.cc:180-252)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVO_LIB_PATH") or os.path.join(_HERE, "libsvo_hip.so")   # (SVO_LIB_PATH: A/B runs against another build of the library)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

TRACK_DTYPE = np.dtype([("Tcw", "<f4", (16,)), ("frame_id", "<i4"), ("n_kp", "<i4"),
                        ("n_stereo", "<i4"), ("n_match_pass1", "<i4"), ("n_match_pass2", "<i4"),
                        ("n_pnp_inliers", "<i4"), ("n_lm_edges", "<i4"), ("n_new_mappoints", "<i4"),
                        ("n_local_map", "<i4"), ("lm_iterations", "<i4"), ("reserved", "<i4", (2,))])

# svo_track_debug (include/svo.h): parity probe record of one tracked frame
TRACK_DEBUG_DTYPE = np.dtype([("match_gid", "<i4", (512,)), ("new_gid", "<i4", (512,)), ("frame_id", "<i4"),
                              ("pnp_best", "<i4"), ("pnp_iterations", "<i4"), ("pnp_inliers", "<i4"), ("pnp_ok", "<i4"),
                              ("active_rows", "<i4", (2,)), ("rounds", "<i4", (2,)), ("resolve_us", "<i4"),
                              ("rt", "<i8", (6,)), ("T_pnp", "<f8", (16,))])

# Every symbol include/svo.h declares (checked by tests/test_abi.py without a GPU).
ABI_SYMBOLS = [
    "svo_abi_version", "svo_strerror", "svo_last_error", "svo_create", "svo_destroy", "svo_sync",
    "svo_set_option",
    "svo_stream", "svo_orb_geometry", "svo_orb_extract", "svo_debug_pyramid_level",
    "svo_debug_fast_corners", "svo_stereo_frame", "svo_stereo_frame_ex", "svo_disp2depth",
    "svo_unproject", "svo_descriptor_distance", "svo_hamming_argmin", "svo_match_greedy",
    "svo_match_greedy_gated",
    "svo_bf_match", "svo_pnp_ransac", "svo_debug_epnp5", "svo_pose_opt", "svo_track_reset", "svo_track_frame",
    "svo_debug_track_matches", "svo_debug_track_gate", "svo_debug_track_pnp", "svo_debug_track_frames", "svo_fundamental_8point",
    "svo_frontend_batch_dev", "svo_track_batch_dev", "svo_profile_enable", "svo_profile_reset",
    "svo_profile_get",
    "svo_elas_default_params", "svo_elas_process", "svo_elas_process_ex", "svo_elas_delaunay",
    "svo_ctmf", "svo_track_multi_reset", "svo_track_multi_step_dev", "svo_track_tail_dev", "svo_track_overflowed", "svo_track_epnp_fallbacks", "svo_debug_stream_probe", "svo_debug_stream_pipes", "svo_track_sharded_dev", "svo_elas_batch_dev", "svo_msa_init", "svo_msa_tree", "svo_msa_tree_dp", "svo_msa_wta", "svo_msa_lrcheck", "svo_msa_solve", "svo_msa_batch_dev",
    "svo_create_ex", "svo_stream_mode", "svo_track_batch_host", "svo_track_sharded_host", "svo_frontend_batch_host",
]

# svo_create_ex flags (include/svo.h)
CREATE_POOLED_STREAMS = 1
CREATE_TAIL_ALL_CUS = 2


class SvoError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("bf", C.c_float)]


class BoxesDev(C.Structure):
    """svo_boxes_dev: per-frame detection boxes as HBM arrays (device pointers)."""
    _fields_ = [("boxes", C.c_void_p), ("n", C.c_void_p), ("stride", C.c_int32)]


def boxes_dev(d_boxes, d_n, stride):
    """svo_boxes_dev from two device pointers (e.g. torch int32 tensors' data_ptr()): boxes (F x stride x 4), n (F)."""
    return BoxesDev(C.c_void_p(int(d_boxes)), C.c_void_p(int(d_n)), int(stride))


def _bx(b):
    return None if b is None else C.byref(b)


class BoxesHost(C.Structure):
    """svo_boxes_host: the same as HOST arrays (the host-fed entries)."""
    _fields_ = [("boxes", C.c_void_p), ("n", C.c_void_p), ("stride", C.c_int32)]


def boxes_host(boxes, n):
    """svo_boxes_host from two numpy int32 arrays: boxes (F x stride x 4), n (F).  Keeps them alive in the returned object."""
    b = np.ascontiguousarray(boxes, np.int32)
    c = np.ascontiguousarray(n, np.int32)
    r = BoxesHost(b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), int(b.shape[1]))
    r._keep = (b, c)
    return r


class LmStats(C.Structure):
    _fields_ = [("n_edges", C.c_int32), ("iterations", C.c_int32), ("trials_total", C.c_int32),
                ("terminated", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double)]


class PnpStats(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("n_inliers", C.c_int32), ("best_hypothesis", C.c_int32),
                ("ok", C.c_int32), ("iterations", C.c_int32)]


# KITTI intrinsics of the reference's settings files (Stereo/KITTI00-02.yaml:8-11,25 and
# Stereo/KITTI04-12.yaml:8-11,25) - the only five keys Tracking::Tracking reads.
KITTI_00_02 = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448)
KITTI_04_12 = dict(fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104, bf=379.8145)

_lib = None


def load_library():
    """dlopen libsvo_hip.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvoError("HIP extension missing: %s (run `make -C %s` or __graft_entry__.build())"
                           % (LIB_PATH, _HERE))
        try:
            # torch wheels bundle their own libamdhip64; two HIP runtimes in one process cannot
            # both see the GPU.  Loading torch first makes its runtime the single one (our
            # library's libamdhip64.so.7 dependency then resolves to it).
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.svo_strerror.restype = C.c_char_p
        lib.svo_last_error.restype = C.c_char_p
        lib.svo_last_error.argtypes = [C.c_void_p]
        lib.svo_stream.restype = C.c_void_p
        lib.svo_stream.argtypes = [C.c_void_p]
        lib.svo_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.svo_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        lib.svo_stream_mode.argtypes = [C.c_void_p]
        lib.svo_destroy.argtypes = [C.c_void_p]
        lib.svo_destroy.restype = None
        _lib = lib
    return _lib


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


class Svo:
    """One tracker context bound to one GPU (wraps svo_ctx)."""

    def __init__(self, W, H, device=0, max_kp=500, max_batch=1, flags=None):
        """flags: None = svo_create (default stream mode, SVO_POOLED_QUEUES honoured), else svo_create_ex with CREATE_* bits."""
        self.lib = load_library()
        self.W, self.H, self.max_kp, self.max_batch = int(W), int(H), int(max_kp), int(max_batch)
        h = C.c_void_p()
        if flags is None:
            rc = self.lib.svo_create(C.byref(h), int(device), self.W, self.H, self.max_kp, self.max_batch)
        else:
            rc = self.lib.svo_create_ex(C.byref(h), int(device), self.W, self.H, self.max_kp, self.max_batch, int(flags))
        if rc != 0:
            raise SvoError("svo_create failed: %s" % self.lib.svo_strerror(rc).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.svo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SvoError("%s: %s" % (self.lib.svo_strerror(rc).decode(),
                                       self.lib.svo_last_error(self.h).decode()))

    @staticmethod
    def camera(fx, fy, cx, cy, bf):
        return Camera(fx, fy, cx, cy, bf)

    def set_option(self, key, value):
        self._chk(self.lib.svo_set_option(self.h, key.encode(), int(value)))

    def sync(self):
        self._chk(self.lib.svo_sync(self.h))

    @property
    def stream(self):
        return self.lib.svo_stream(self.h)

    def geometry(self):
        w = np.zeros(8, np.int32); h = np.zeros(8, np.int32)
        s = np.zeros(8, np.float32); q = np.zeros(8, np.int32)
        self._chk(self.lib.svo_orb_geometry(self.h, _p(w), _p(h), _p(s), _p(q)))
        return w, h, s, q

    # ---- frame::featuredetect ------------------------------------------------------
    def orb_extract(self, gray):
        gray = _u8(gray)
        assert gray.shape == (self.H, self.W)
        kp = np.zeros(self.max_kp, KP_DTYPE); desc = np.zeros((self.max_kp, 32), np.uint8)
        n = C.c_int32(0)
        self._chk(self.lib.svo_orb_extract(self.h, _p(gray), self.W, _p(kp), _p(desc), C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def debug_pyramid_level(self, slot, level):
        w, h, _, _ = self.geometry()
        out = np.zeros((h[level], w[level]), np.uint8)
        self._chk(self.lib.svo_debug_pyramid_level(self.h, slot, level, _p(out)))
        return out

    def debug_fast_corners(self, slot, level):
        w, h, _, _ = self.geometry()
        cap = (int(w[level]) // 2 + 1) * (int(h[level]) // 2 + 1)
        out = np.zeros((cap, 3), np.int32); n = C.c_int32(0)
        self._chk(self.lib.svo_debug_fast_corners(self.h, slot, level, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    # ---- stereo association + depth -------------------------------------------------
    def stereo_frame(self, grayL, grayR, cam):
        grayL = _u8(grayL); grayR = _u8(grayR)
        assert grayL.shape == (self.H, self.W) and grayR.shape == (self.H, self.W)
        K = self.max_kp
        kpL = np.zeros(K, KP_DTYPE); dL = np.zeros((K, 32), np.uint8)
        kpR = np.zeros(K, KP_DTYPE); dR = np.zeros((K, 32), np.uint8)
        uR = np.zeros(K, np.float32); depth = np.zeros(K, np.float32)
        nL = C.c_int32(0); nR = C.c_int32(0)
        self._chk(self.lib.svo_stereo_frame_ex(self.h, _p(grayL), self.W, _p(grayR), self.W,
                                               C.byref(cam), _p(kpL), _p(dL), C.byref(nL), _p(uR),
                                               _p(depth), _p(kpR), _p(dR), C.byref(nR)))
        nl, nr = nL.value, nR.value
        return dict(kpL=kpL[:nl].copy(), dL=dL[:nl].copy(), uR=uR[:nl].copy(),
                    depth=depth[:nl].copy(), kpR=kpR[:nr].copy(), dR=dR[:nr].copy())

    def disp2depth(self, disp, bf):
        disp = np.ascontiguousarray(disp, np.float32); out = np.zeros_like(disp)
        self._chk(self.lib.svo_disp2depth(self.h, _p(disp), disp.size, C.c_float(bf), _p(out)))
        return out

    def unproject(self, uvz, cam, Rwc, twc):
        uvz = np.ascontiguousarray(uvz, np.float32).reshape(-1, 3)
        Rwc = np.ascontiguousarray(Rwc, np.float32).reshape(9)
        twc = np.ascontiguousarray(twc, np.float32).reshape(3)
        out = np.zeros_like(uvz)
        self._chk(self.lib.svo_unproject(self.h, _p(uvz), len(uvz), C.byref(cam), _p(Rwc), _p(twc),
                                         _p(out)))
        return out

    # ---- pnpmatch ---------------------------------------------------------------------
    def descriptor_distance(self, a, b):
        a = _u8(a).reshape(-1, 32); b = _u8(b).reshape(-1, 32)
        out = np.zeros(len(a), np.int32)
        self._chk(self.lib.svo_descriptor_distance(self.h, _p(a), _p(b), len(a), _p(out)))
        return out

    def hamming_argmin(self, q, t, t_mask=None):
        q = _u8(q).reshape(-1, 32); t = _u8(t).reshape(-1, 32)
        M, N = len(q), len(t)
        bi = np.zeros(M, np.int32); b = np.zeros(M, np.int32); s = np.zeros(M, np.int32)
        m = None if t_mask is None else _u8(t_mask)
        self._chk(self.lib.svo_hamming_argmin(self.h, _p(q), M, _p(t), N, _p(m), _p(bi), _p(b), _p(s)))
        return bi, b, s

    def match_greedy(self, q, t, assigned, max_dist, ratio, q_skip=None):
        q = _u8(q).reshape(-1, 32); t = _u8(t).reshape(-1, 32)
        M, N = len(q), len(t)
        assigned = _u8(assigned).copy()
        sk = None if q_skip is None else _u8(q_skip)
        bi = np.zeros(M, np.int32); b = np.zeros(M, np.int32); s = np.zeros(M, np.int32)
        acc = np.zeros(M, np.uint8)
        self._chk(self.lib.svo_match_greedy(self.h, _p(q), _p(sk), M, _p(t), N, _p(assigned),
                                            int(max_dist), C.c_float(ratio), _p(bi), _p(b), _p(s),
                                            _p(acc)))
        return bi, b, s, acc, assigned

    def bf_match(self, q, t):
        q = _u8(q).reshape(-1, 32); t = _u8(t).reshape(-1, 32)
        M, N = len(q), len(t)
        ti = np.zeros(M, np.int32); d = np.zeros(M, np.int32); keep = np.zeros(M, np.uint8)
        self._chk(self.lib.svo_bf_match(self.h, _p(q), M, _p(t), N, _p(ti), _p(d), _p(keep)))
        return ti, d, keep

    def pnp_ransac(self, Xw, obs, K, T_fallback, rng_state=0):
        """cv::solvePnPRansac(..., false, 100, 8.0, 0.99) (svo_pnp_ransac); rng_state 0 = OpenCV's (uint64)-1."""
        Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        Tp = np.ascontiguousarray(T_fallback, np.float64).reshape(16)
        T = np.zeros(16); mask = np.zeros(max(len(Xw), 1), np.uint8); st = PnpStats()
        self._chk(self.lib.svo_pnp_ransac(self.h, _p(Xw), _p(obs), len(Xw), _p(K), _p(Tp),
                                          C.c_uint64(rng_state), _p(T), _p(mask), C.byref(st)))
        return T.reshape(4, 4), mask[:len(Xw)], st

    def debug_epnp5(self, Xw5, uv5, K):
        Xw5 = np.ascontiguousarray(Xw5, np.float64).reshape(15); uv5 = np.ascontiguousarray(uv5, np.float64).reshape(10)
        K = np.ascontiguousarray(K, np.float64); R = np.zeros(9); t = np.zeros(3); rep = np.zeros(3)
        self._chk(self.lib.svo_debug_epnp5(self.h, _p(Xw5), _p(uv5), _p(K), _p(R), _p(t), _p(rep)))
        return R.reshape(3, 3), t, rep

    # ---- Optimizer::PoseOptimization ------------------------------------------------------
    def pose_opt(self, Xw, obs, K, T):
        Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        T = np.ascontiguousarray(T, np.float64).reshape(16).copy()
        st = LmStats()
        self._chk(self.lib.svo_pose_opt(self.h, _p(Xw), _p(obs), len(Xw), _p(K), _p(T), C.byref(st)))
        return T.reshape(4, 4), st

    # ---- Tracking::Track -------------------------------------------------------------------
    def track_reset(self, cam):
        self._chk(self.lib.svo_track_reset(self.h, C.byref(cam)))

    def track_frame(self, grayL, grayR, timestamp=0.0, boxes=None):
        grayL = _u8(grayL); grayR = _u8(grayR)
        res = np.zeros(1, TRACK_DTYPE)
        bx = None if boxes is None or len(boxes) == 0 else np.ascontiguousarray(boxes, np.int32)
        nb = 0 if bx is None else len(bx)
        self._chk(self.lib.svo_track_frame(self.h, _p(grayL), self.W, _p(grayR), self.W,
                                           C.c_double(timestamp), _p(bx), nb, _p(res)))
        return res[0]

    def debug_track_gate(self):
        F = np.zeros(9); n = C.c_int32(0)
        self._chk(self.lib.svo_debug_track_gate(self.h, _p(F), C.byref(n)))
        return F.reshape(3, 3), n.value

    def fundamental_8point(self, pts1, pts2):
        """cv::findFundamentalMat(pts1, pts2, CV_FM_8POINT) on the device (svo_fundamental_8point)."""
        pts1 = np.ascontiguousarray(pts1, np.float64).reshape(-1, 2)
        pts2 = np.ascontiguousarray(pts2, np.float64).reshape(-1, 2)
        F = np.zeros(9)
        self._chk(self.lib.svo_fundamental_8point(self.h, _p(pts1), _p(pts2), len(pts1), _p(F)))
        return F.reshape(3, 3)

    def debug_track_frames(self, first, n):
        """svo_debug_track_frames: TRACK_DEBUG_DTYPE records of n frames of the last device-resident call."""
        out = np.zeros(n, TRACK_DEBUG_DTYPE)
        assert TRACK_DEBUG_DTYPE.itemsize == 4096 + 10 * 4 + 48 + 128
        self._chk(self.lib.svo_debug_track_frames(self.h, int(first), int(n), _p(out)))
        return out

    def debug_track_matches(self):
        out = np.zeros(self.max_kp, np.int32)
        self._chk(self.lib.svo_debug_track_matches(self.h, _p(out)))
        return out

    # ---- throughput mode (device pointers, e.g. torch tensors' data_ptr()) --------------------
    def frontend_batch_dev(self, d_grayL, d_grayR, stride, B, cam, d_kpL=None, d_descL=None,
                           d_nL=None, d_uR=None, d_depth=None):
        self._chk(self.lib.svo_frontend_batch_dev(self.h, _p(d_grayL), _p(d_grayR), int(stride),
                                                  int(B), C.byref(cam), _p(d_kpL), _p(d_descL),
                                                  _p(d_nL), _p(d_uR), _p(d_depth)))

    def track_batch_dev(self, d_grayL, d_grayR, stride, B, d_results, boxes=None):
        """boxes: a BoxesDev (boxes_dev(...)) with the frames' offline detection boxes in HBM, or None."""
        self._chk(self.lib.svo_track_batch_dev(self.h, _p(d_grayL), _p(d_grayR), int(stride), int(B), _bx(boxes),
                                               _p(d_results)))

    def stream_mode(self):
        """svo_stream_mode: bit 0 = four dedicated hardware queues, bit 1 = the tail's queues keep off the front end's CUs."""
        return int(self.lib.svo_stream_mode(self.h))

    def track_batch_host(self, grayL, grayR, stride, B, results, boxes=None):
        """svo_track_batch_host: B consecutive frames that start in HOST memory (pointers or numpy arrays; pinned memory is copied
        where it lies, pageable memory is staged inside the call); results: host array of TRACK_DTYPE records (complete after
        sync()).  boxes: a BoxesHost or None."""
        self._chk(self.lib.svo_track_batch_host(self.h, _p(grayL), _p(grayR), int(stride), int(B), _bx(boxes), _p(results)))

    @staticmethod
    def track_sharded_host(ctxs, grayL, grayR, stride, B, results, boxes=None):
        """svo_track_sharded_host: ONE sequence in host memory, pair k uploaded to and extracted on ctxs[k % G], tail on ctxs[0]."""
        G = len(ctxs)
        hs = (C.c_void_p * G)(*[c.h for c in ctxs])
        ctxs[0]._chk(ctxs[0].lib.svo_track_sharded_host(hs, G, _p(grayL), _p(grayR), int(stride), int(B), _bx(boxes), _p(results)))

    def frontend_batch_host(self, grayL, grayR, stride, B, cam, kpL=None, descL=None, nL=None, uR=None, depth=None):
        """svo_frontend_batch_host: the stateless front end host to host, pipelined; outputs complete after sync()."""
        self._chk(self.lib.svo_frontend_batch_host(self.h, _p(grayL), _p(grayR), int(stride), int(B), C.byref(cam), _p(kpL),
                                                   _p(descL), _p(nL), _p(uR), _p(depth)))

    def track_tail_dev(self, d_kp, d_desc, d_n, d_depth, kp_stride, B, d_results, boxes=None):
        """svo_track_tail_dev: the ordered tail over front-end results already in HBM (device pointers)."""
        self._chk(self.lib.svo_track_tail_dev(self.h, _p(d_kp), _p(d_desc), _p(d_n), _p(d_depth), int(kp_stride),
                                              int(B), _bx(boxes), _p(d_results)))

    @staticmethod
    def track_sharded_dev(ctxs, d_grayL, d_grayR, stride, B, d_results, boxes=None):
        """svo_track_sharded_dev: ONE sequence, pair k on ctxs[k % G], ordered tail on ctxs[0]."""
        G = len(ctxs)
        hs = (C.c_void_p * G)(*[c.h for c in ctxs])
        pl = (C.c_void_p * G)(*[C.c_void_p(int(p)) for p in d_grayL])
        pr = (C.c_void_p * G)(*[C.c_void_p(int(p)) for p in d_grayR])
        ctxs[0]._chk(ctxs[0].lib.svo_track_sharded_dev(hs, G, pl, pr, int(stride), int(B), _bx(boxes), _p(d_results)))

    def track_overflowed(self):
        f = C.c_int32(0)
        self._chk(self.lib.svo_track_overflowed(self.h, C.byref(f)))
        return f.value

    def track_epnp_fallbacks(self):
        n = C.c_int64(0)
        self._chk(self.lib.svo_track_epnp_fallbacks(self.h, C.byref(n)))
        return n.value

    def debug_stream_probe(self):
        """(candidates tried, polls of the probe's waiting kernel) of the index chain's stream; see include/svo.h."""
        out = (C.c_int32 * 2)()
        self._chk(self.lib.svo_debug_stream_probe(self.h, out))
        return int(out[0]), int(out[1])

    def debug_stream_pipes(self):
        """Measured: how late a grid on the (pose, index) stream is beside queued filling grids on the front end's stream, then the
        dense stage's - 100 + 100 x rounds, -1 where a stream does not exist; see include/svo.h."""
        out = (C.c_int32 * 4)()
        self._chk(self.lib.svo_debug_stream_pipes(self.h, out))
        return [int(v) for v in out]

    def track_multi_reset(self, n_seq, cam):
        self._chk(self.lib.svo_track_multi_reset(self.h, int(n_seq), C.byref(cam)))

    def track_multi_step_dev(self, d_grayL, d_grayR, stride, n_seq, d_results, boxes=None):
        self._chk(self.lib.svo_track_multi_step_dev(self.h, C.c_void_p(d_grayL), C.c_void_p(d_grayR), int(stride),
                                                     int(n_seq), _bx(boxes), C.c_void_p(d_results)))

    def profile_enable(self, on=True):
        self._chk(self.lib.svo_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.lib.svo_profile_reset(self.h))

    def profile(self):
        out, i = {}, 0
        name = C.create_string_buffer(64); ms = C.c_double(0); n = C.c_int64(0)
        while self.lib.svo_profile_get(self.h, i, name, 64, C.byref(ms), C.byref(n)) == 0:
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    # ---- dense ELAS stereo (include/svo.h: svo_elas_*; replaces Thirdparty/libelas Elas::process) ----
    def elas_process(self, grayL, grayR, params=None, taps=False, tri1=None, tri2=None):
        """D1, D2 (float32 H x W, negative = invalid).  With taps=True returns a dict of every
        intermediate instead (keys as in svo_elas_taps, plus "D1"/"D2")."""
        gl, gr = _u8(grayL), _u8(grayR)
        H, W = gl.shape
        params = params or elas_default_params(0)
        dims = (C.c_int32 * 3)(W, H, W)
        Hd, Wd = (H // 2, W // 2) if params.subsampling else (H, W)
        D1 = np.zeros((Hd, Wd), np.float32); D2 = np.zeros((Hd, Wd), np.float32)
        if not taps and tri1 is None and tri2 is None:
            self._chk(self.lib.svo_elas_process(self.h, _p(gl), _p(gr), _p(D1), _p(D2), dims, C.byref(params)))
            return D1, D2
        cap_sp = (W // params.candidate_stepsize + 2) * (H // params.candidate_stepsize + 2) + 8
        cap_tri = 2 * cap_sp + 16
        gw = -(-W // params.grid_size); gh = -(-H // params.grid_size)
        o = dict(desc1=np.zeros((H, W, 16), np.uint8), desc2=np.zeros((H, W, 16), np.uint8),
                 support=np.zeros((cap_sp, 3), np.int32),
                 tri1=np.zeros((cap_tri, 3), np.int32), tri2=np.zeros((cap_tri, 3), np.int32),
                 planes1=np.zeros((cap_tri, 6), np.float32), planes2=np.zeros((cap_tri, 6), np.float32),
                 grid1=np.zeros((gh, gw, params.disp_max + 2), np.int32),
                 grid2=np.zeros((gh, gw, params.disp_max + 2), np.int32))
        for k in ("raw", "lr", "seg", "gap", "mean"):
            o["D1_" + k] = np.zeros((Hd, Wd), np.float32); o["D2_" + k] = np.zeros((Hd, Wd), np.float32)
        t = ElasTaps()
        for k, a in o.items():
            setattr(t, k, a.ctypes.data)
        t.cap_support = cap_sp; t.cap_tri = cap_tri
        keep = []
        for name, tri in (("tri1_in", tri1), ("tri2_in", tri2)):
            if tri is not None:
                a = np.ascontiguousarray(tri, np.int32); keep.append(a)
                setattr(t, name, a.ctypes.data); setattr(t, "n_" + name, len(a))
        self._chk(self.lib.svo_elas_process_ex(self.h, _p(gl), _p(gr), _p(D1), _p(D2), dims, C.byref(params),
                                               C.byref(t)))
        o["support"] = o["support"][:t.n_support].copy()
        for s in ("1", "2"):
            n = getattr(t, "n_tri" + s)
            o["tri" + s] = o["tri" + s][:n].copy(); o["planes" + s] = o["planes" + s][:n].copy()
        o["D1"] = D1; o["D2"] = D2
        return o

    def elas_batch_dev(self, d_L, d_R, stride, W, H, B, d_D1, d_D2, params=None):
        """B device-resident pairs -> B pairs of device-resident maps; returns the per-pair `produced` flags."""
        params = params or elas_default_params(0)
        produced = np.zeros(B, np.int32)
        self._chk(self.lib.svo_elas_batch_dev(self.h, C.c_void_p(d_L), C.c_void_p(d_R), int(stride), int(W), int(H),
                                              int(B), C.byref(params), C.c_void_p(d_D1), C.c_void_p(d_D2), _p(produced)))
        return produced

    def msa_batch_dev(self, d_L, d_R, stride, W, H, B, d_disp, d=48):
        """B device-resident gray pairs -> B device-resident float disparity maps (MSA::solve, scale 1)."""
        self._chk(self.lib.svo_msa_batch_dev(self.h, C.c_void_p(d_L), C.c_void_p(d_R), int(stride), int(W), int(H), int(B),
                                             int(d), C.c_void_p(d_disp)))

    def msa_init(self, bgrL, bgrR, disp=49):
        """MSA::init on two H x W x 3 uint8 images: cost volumes, median images, gradients (dict)."""
        a, b = _u8(bgrL), _u8(bgrR)
        H, W = a.shape[:2]
        o = dict(costL=np.zeros((H, W, disp), np.float32), costR=np.zeros((H, W, disp), np.float32),
                 m3L=np.zeros_like(a), m3R=np.zeros_like(a),
                 r_graL=np.zeros((H, W)), c_graL=np.zeros((H, W)), r_graR=np.zeros((H, W)), c_graR=np.zeros((H, W)))
        self._chk(self.lib.svo_msa_init(self.h, _p(a), _p(b), W, H, 3 * W, int(disp), _p(o["costL"]), _p(o["costR"]),
                                        _p(o["m3L"]), _p(o["m3R"]), _p(o["r_graL"]), _p(o["c_graL"]), _p(o["r_graR"]),
                                        _p(o["c_graR"])))
        return o

    def msa_tree_dp(self, cost, seq, child_ptr, child, child_c, root, o=0.1):
        cost = np.ascontiguousarray(cost, np.float32)
        N, D = cost.shape
        A = np.zeros_like(cost)
        self.lib.svo_msa_tree_dp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        self._chk(self.lib.svo_msa_tree_dp(self.h, _p(cost), N, D, _p(np.ascontiguousarray(seq, np.int32)),
                                           _p(np.ascontiguousarray(child_ptr, np.int32)),
                                           _p(np.ascontiguousarray(child, np.int32)),
                                           _p(np.ascontiguousarray(child_c, np.uint8)), int(root), float(o), _p(A)))
        return A

    def msa_wta(self, costA, H, W):
        costA = np.ascontiguousarray(costA, np.float32)
        out = np.zeros((H, W), np.uint8)
        self._chk(self.lib.svo_msa_wta(self.h, _p(costA), W, H, costA.shape[-1], _p(out)))
        return out

    def msa_lrcheck(self, d1, d2, D):
        d1, d2 = _u8(d1), _u8(d2)
        H, W = d1.shape
        cost = np.zeros((H, W, D), np.float32); mask = np.zeros((H, W), np.uint8)
        self._chk(self.lib.svo_msa_lrcheck(self.h, _p(d1), _p(d2), W, H, int(D), _p(cost), _p(mask)))
        return cost, mask

    def msa_solve(self, bgrL, bgrR, d=48, scale=1):
        """MSA::solve: H x W uint8 disparity image (left reference)."""
        a, b = _u8(bgrL), _u8(bgrR)
        H, W = a.shape[:2]
        out = np.zeros((H, W), np.uint8)
        self._chk(self.lib.svo_msa_solve(self.h, _p(a), _p(b), W, H, 3 * W, int(d), int(scale), _p(out)))
        return out

    def ctmf(self, img, r):
        """Median filter of Thirdparty/MB/ctmf.c on an H x W or H x W x C uint8 image."""
        a = _u8(img)
        cn = 1 if a.ndim == 2 else a.shape[2]
        H, W = a.shape[:2]
        out = np.zeros_like(a)
        self._chk(self.lib.svo_ctmf(self.h, _p(a), _p(out), W, H, W * cn, W * cn, int(r), cn))
        return out


class ElasParams(C.Structure):
    """svo_elas_params == Elas::parameters (Thirdparty/libelas/src/elas.h:60-83), bools as int32."""
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


class ElasTaps(C.Structure):
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


def elas_default_params(setting=0):
    """Elas::parameters(setting): 0 = ROBOTICS (the reference's default), 1 = MIDDLEBURY."""
    p = ElasParams()
    rc = load_library().svo_elas_default_params(int(setting), C.byref(p))
    if rc != 0:
        raise SvoError("svo_elas_default_params failed")
    return p


def msa_tree(m_img3, r_gra, c_gra):
    """Host-side MSA aggregation tree of one image (needs no GPU): (root, seq, child_ptr, child, child_c)."""
    a = _u8(m_img3)
    H, W = a.shape[:2]
    N = H * W
    seq = np.zeros(N, np.int32); cp = np.zeros(N + 1, np.int32); ch = np.zeros(N, np.int32); cc = np.zeros(N, np.uint8)
    root = C.c_int32(-1)
    rc = load_library().svo_msa_tree(_p(a), _p(np.ascontiguousarray(r_gra, np.float64)),
                                     _p(np.ascontiguousarray(c_gra, np.float64)), W, H, _p(seq), _p(cp), _p(ch), _p(cc),
                                     C.byref(root))
    if rc != 0:
        raise SvoError("svo_msa_tree failed (%d)" % rc)
    return root.value, seq, cp, ch[:N - 1].copy(), cc[:N - 1].copy()


def elas_delaunay(xy):
    """Host-side Delaunay triangulation of n (x, y) integer points (needs no GPU)."""
    xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
    cap = 4 * len(xy) + 16
    tri = np.zeros((cap, 3), np.int32)
    n = C.c_int32(0)
    rc = load_library().svo_elas_delaunay(_p(xy), len(xy), _p(tri), cap, C.byref(n))
    if rc != 0:
        raise SvoError("svo_elas_delaunay failed (%d)" % rc)
    return tri[:n.value].copy()
