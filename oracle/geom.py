"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/_build/liboracle_c.so (oracle/oracle_geom.c) plus numpy
restatements of the small host-side geometry of the reference.  Never imported by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle_c.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_logdepth_sum.restype = C.c_double
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def rope2d(tokens, pos, base=100.0, fwd=1.0):
    """tokens [B,N,H,D] fp32 -> rotated copy (CUDA-kernel operation order, kernels.cu:43-54)."""
    t = _f(tokens).copy()
    p = np.ascontiguousarray(pos, dtype=np.int64)
    B, N, H, D = t.shape
    lib().oracle_rope2d_f32(_ptr(t), _ptr(p, C.c_int64), B, N, H, D, C.c_float(base), C.c_float(fwd))
    return t


def w2c_rows(c2w):
    """inverse of rigid/affine c2w [B,4,4] in float64, returned as fp32 [B,12] (top 3x4, row-major)."""
    c2w = np.asarray(c2w, dtype=np.float64).reshape(-1, 4, 4)
    inv = np.linalg.inv(c2w)
    return np.ascontiguousarray(inv[:, :3, :].reshape(-1, 12), dtype=np.float32)


def overlap_fwd(pm, w2c12, K4, W, H, clamp_z=True):
    pm = _f(pm).reshape(-1, 3)
    w = _f(w2c12).reshape(-1, 12)
    k = _f(K4)
    out = np.zeros(w.shape[0], np.int32)
    lib().oracle_overlap_fwd_ex(_ptr(pm), pm.shape[0], _ptr(w), w.shape[0], _ptr(k), int(W), int(H), int(clamp_z), _ptr(out, C.c_int32))
    return out


def overlap_bwd(pms, w2c12, K4, W, H):
    pms = _f(pms)
    B = pms.shape[0]
    N = pms[0].size // 3
    w = _f(w2c12).reshape(12)
    k = _f(K4)
    out = np.zeros(B, np.int32)
    lib().oracle_overlap_bwd(_ptr(pms), B, N, _ptr(w), _ptr(k), int(W), int(H), _ptr(out, C.c_int32))
    return out


def align_view(pts, conf, P12, s, ds):
    pts, conf = _f(pts), _f(conf)
    H, W = conf.shape
    pm = np.zeros((H // ds, W // ds, 3), np.float32)
    cd = np.zeros((H // ds, W // ds), np.float32)
    dp = np.zeros((H, W), np.float32)
    P = _f(P12).reshape(12)
    lib().oracle_align_view(_ptr(pts), _ptr(conf), H, W, _ptr(P), C.c_float(s), int(ds), _ptr(pm), _ptr(cd), _ptr(dp))
    return pm, cd, dp


def logdepth_sum(prev_depth, pts):
    a, b = _f(prev_depth), _f(pts)
    return float(lib().oracle_logdepth_sum(_ptr(a), _ptr(b), a.size))


def patch_overlap_ratio(feat0, feat1, thr=0.7):
    """hislam2/util/utils.py:726-736 in float64 (the decision both fp32 evaluations approximate)."""
    a = np.asarray(feat0, np.float64)[1:]
    b = np.asarray(feat1, np.float64)[1:]
    a = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
    b = b / np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-12)
    mx = (a @ b.T).max(axis=1)
    return float((mx > thr).mean()), mx


def resize_linear_u8(img, H1, W1):
    """cv2.resize(img, (W1, H1)) (INTER_LINEAR, u8 HWC) per the published OpenCV algorithm -- PARITY UNPINNED (no cv2 here)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H0, W0, Cc = img.shape
    out = np.empty((H1, W1, Cc), np.uint8)
    lib().oracle_resize_linear_u8(_ptr(img, C.c_ubyte), H0, W0, Cc, _ptr(out, C.c_ubyte), H1, W1)
    return out


def undistort_map(K4, dist, H, W):
    """cv2.initUndistortRectifyMap(K, dist, None, K, (W, H), CV_16SC2) restated pixel by pixel (plain loops, float64): source
    coordinates of every destination pixel in 1/32 pixel.  PARITY UNPINNED (no cv2 here)."""
    fx, fy, cx, cy = [float(v) for v in K4]
    d = [float(v) for v in dist] + [0.0] * (8 - len(dist))
    k1, k2, p1, p2, k3, k4, k5, k6 = d
    ix = np.zeros((H, W), np.int32)
    iy = np.zeros((H, W), np.int32)
    for i in range(H):
        y = (i - cy) / fy
        for j in range(W):
            x = (j - cx) / fx
            r2 = x * x + y * y
            kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
            xd = x * kr + p1 * 2 * x * y + p2 * (r2 + 2 * x * x)
            yd = y * kr + p1 * (r2 + 2 * y * y) + p2 * 2 * x * y
            ix[i, j] = int(round(32.0 * (fx * xd + cx)))          # Python round: ties to even, like cvRound
            iy[i, j] = int(round(32.0 * (fy * yd + cy)))
    return ix, iy


def remap_linear_u8(img, ix, iy):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, Cc = img.shape
    ix = np.ascontiguousarray(ix, dtype=np.int32)
    iy = np.ascontiguousarray(iy, dtype=np.int32)
    Ho, Wo = ix.shape
    out = np.empty((Ho, Wo, Cc), np.uint8)
    lib().oracle_remap_linear_u8(_ptr(img, C.c_ubyte), H, W, Cc, _ptr(ix, C.c_int32), _ptr(iy, C.c_int32), _ptr(out, C.c_ubyte), Ho, Wo)
    return out
