"""Image-sequence front end with the reference's generator contract (/root/reference/demo_s.py:48-94 `mono_stream`,
:97-113 `save_trajectory`): same file ordering (natural sort), same intrinsics arithmetic, same two resizes -- but the
frame is uploaded once and both resizes run on the GPU (`cut3r_resize_linear_u8`: OpenCV's INTER_LINEAR fixed-point
bilinear, parity unpinned because cv2 is not available here; image decoding is PIL instead of cv2.imread, which can differ
from libjpeg-turbo-in-OpenCV by +-1 LSB on JPEGs).

Not supported (raises): lens undistortion (`cv2.undistort`, demo_s.py:63-64) -- Replica/BASELINE sequences carry 4-value
calibrations.
"""
from __future__ import annotations

import os
import re

import numpy as np
import torch

from . import ops

_NUM = re.compile(r"(\d+)")


def natural_key(name: str):
    """natsort-style key: digit runs compare as integers ("frame10" after "frame9")"""
    return [int(tok) if tok.isdigit() else tok.lower() for tok in _NUM.split(name)]


def natsorted(names):
    return sorted(names, key=natural_key)


def tracking_size(h0: int, w0: int):
    """demo_s.py:69-71: width 512, height scaled and floored to a multiple of 16"""
    return int((512 / w0 * h0) // 16) * 16, 512


def mapping_size(h0: int, w0: int):
    """demo_s.py:81-82: width 512, height floored to an even number"""
    return int(512 / w0 * h0) // 2 * 2, 512


def load_calib(path: str) -> np.ndarray:
    return np.loadtxt(path, delimiter=" ")


def undistort_map(K4, dist, H: int, W: int):
    """initUndistortRectifyMap(K, dist, R = I, newCameraMatrix = K, size, CV_16SC2) as cv2.undistort builds it (demo_s.py:63-64):
    for every destination pixel the source coordinate under the plumb-bob model (k1, k2, p1, p2[, k3[, k4, k5, k6]]), in double
    precision, rounded to 1/32 pixel.  Returns (ix, iy) int32 [H,W] = round(32 u), round(32 v) (ties to even, like cvRound)."""
    fx, fy, cx, cy = [float(v) for v in K4]
    d = np.zeros(8, np.float64)
    dist = np.asarray(dist, np.float64).reshape(-1)
    if dist.size not in (4, 5, 8):
        raise ValueError(f"{dist.size} distortion coefficients: expected 4, 5 or 8 (k1 k2 p1 p2 [k3 [k4 k5 k6]])")
    d[:dist.size] = dist
    k1, k2, p1, p2, k3, k4, k5, k6 = d
    x = (np.arange(W, dtype=np.float64)[None, :] - cx) / fx
    y = (np.arange(H, dtype=np.float64)[:, None] - cy) / fy
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + cx
    v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + cy
    lim = float(2 ** 26)
    ix = np.rint(np.clip(u * 32.0, -lim, lim)).astype(np.int32)
    iy = np.rint(np.clip(v * 32.0, -lim, lim)).astype(np.int32)
    return np.ascontiguousarray(ix), np.ascontiguousarray(iy)


def _decode(path: str) -> np.ndarray:
    from PIL import Image          # RGB order directly (the reference converts cv2's BGR, demo_s.py:60)
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def mono_stream(imagedir, calib, undistort=False, cropborder=0, start=0, length=100000, device="cuda:0"):
    """yields (t, image [1,3,h2,w2] u8, intrinsics [1,4], image_ds [1,3,h1,w1] u8, intrinsics_ds [1,4], is_last);
    the two images are device tensors, the intrinsics float64 host tensors as in the reference."""
    calib = load_calib(calib) if isinstance(calib, (str, os.PathLike)) else np.asarray(calib, np.float64)
    umap = None            # (ix, iy) device maps of cv2.undistort, built on the first frame (they depend on the frame size only)
    image_list = natsorted(os.listdir(imagedir))[start:start + length]
    for t, imfile in enumerate(image_list):
        image = _decode(os.path.join(imagedir, imfile))
        intrinsics = torch.tensor(calib[:4])
        intrinsics_ds = torch.tensor(calib[:4])
        src = None
        if len(calib) > 4 and undistort:                      # demo_s.py:63-64: undistort the full frame, then crop
            raw = torch.from_numpy(np.array(image, dtype=np.uint8, order="C")).to(device, non_blocking=True)
            if umap is None or umap[0].shape != raw.shape[:2]:
                ix, iy = undistort_map(calib[:4], calib[4:], raw.shape[0], raw.shape[1])
                umap = (torch.from_numpy(ix).to(device), torch.from_numpy(iy).to(device))
            src = ops.remap_linear_u8(raw, umap[0], umap[1])
        if cropborder > 0:
            if src is not None:
                src = src[cropborder:-cropborder, cropborder:-cropborder].contiguous()
            else:
                image = image[cropborder:-cropborder, cropborder:-cropborder]
            intrinsics[2:] -= cropborder
            intrinsics_ds[2:] -= cropborder
        if src is not None:
            image = src                                       # only its shape is read below
        h0, w0, _ = image.shape
        if src is None:
            src = torch.from_numpy(np.array(image, dtype=np.uint8, order="C")).to(device, non_blocking=True)
        h1, w1 = tracking_size(h0, w0)
        image_ds = ops.resize_linear_u8(src, h1, w1, chw_out=True)
        intrinsics_ds[0] *= (w1 / w0)
        intrinsics_ds[1] *= (h1 / h0)
        intrinsics_ds[2] *= (w1 / w0)
        intrinsics_ds[3] *= (h1 / h0)
        h2, w2 = mapping_size(h0, w0)
        image_map = ops.resize_linear_u8(src, h2, w2, chw_out=True)
        intrinsics[0] *= (w2 / w0)
        intrinsics[1] *= (h2 / h0)
        intrinsics[2] *= (w2 / w0)
        intrinsics[3] *= (h2 / h0)
        yield (t, image_map[None], intrinsics[None], image_ds[None], intrinsics_ds[None], t == len(image_list) - 1)


def frame_timestamps(imagedir, start=0) -> np.ndarray:
    """demo_s.py:103: the last number in each file name, natural-sorted, as a column"""
    return np.array([float(re.findall(r"[+]?(?:\d*\.\d+|\d+)", x)[-1]) for x in natsorted(os.listdir(imagedir))[start:]])[..., np.newaxis]


def save_trajectory(slam, imagedir, output, start=0, traj_full=None):
    """demo_s.py:97-113: intrinsics.npy, traj_kf.txt ("%.4f" stamp + 7 x "%.7f" c2w pose), optional traj_full.txt"""
    t = slam.keyframes.counter.value - 1
    tstamps = slam.keyframes.tstamp[:t]
    poses_kf = slam.keyframes.pose[:t]
    np.save(os.path.join(output, "intrinsics.npy"), slam.keyframes.intrinsic[0].cpu().numpy())
    tstamps_full = frame_timestamps(imagedir, start)
    tstamps_kf = tstamps_full[tstamps.cpu().numpy().astype(int)]
    ttraj_kf = np.concatenate([tstamps_kf, poses_kf.cpu().numpy()], axis=1)
    np.savetxt(os.path.join(output, "traj_kf.txt"), ttraj_kf, fmt="%.4f %.7f %.7f %.7f %.7f %.7f %.7f %.7f")
    if traj_full is not None:
        np.savetxt(os.path.join(output, "traj_full.txt"), np.concatenate([tstamps_full[:len(traj_full)], traj_full], axis=1))
    return ttraj_kf
