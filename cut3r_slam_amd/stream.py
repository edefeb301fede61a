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


def _decode(path: str) -> np.ndarray:
    from PIL import Image          # RGB order directly (the reference converts cv2's BGR, demo_s.py:60)
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def mono_stream(imagedir, calib, undistort=False, cropborder=0, start=0, length=100000, device="cuda:0"):
    """yields (t, image [1,3,h2,w2] u8, intrinsics [1,4], image_ds [1,3,h1,w1] u8, intrinsics_ds [1,4], is_last);
    the two images are device tensors, the intrinsics float64 host tensors as in the reference."""
    calib = load_calib(calib) if isinstance(calib, (str, os.PathLike)) else np.asarray(calib, np.float64)
    if len(calib) > 4 and undistort:
        raise NotImplementedError("cv2.undistort (demo_s.py:63-64) is not implemented; pass a 4-value calibration")
    image_list = natsorted(os.listdir(imagedir))[start:start + length]
    for t, imfile in enumerate(image_list):
        image = _decode(os.path.join(imagedir, imfile))
        intrinsics = torch.tensor(calib[:4])
        intrinsics_ds = torch.tensor(calib[:4])
        if cropborder > 0:
            image = image[cropborder:-cropborder, cropborder:-cropborder]
            intrinsics[2:] -= cropborder
            intrinsics_ds[2:] -= cropborder
        h0, w0, _ = image.shape
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
