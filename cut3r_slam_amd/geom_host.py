"""Tiny host-side (numpy) geometry of the tracking loop: 7-float poses <-> 4x4 matrices, chaining.

These are O(1)-sized computations that the reference also does on the host after `.cpu()`:
  pose_encoding_to_camera / quaternion_to_matrix   /root/reference/src/dust3r/utils/camera.py:364-420  (quat w,x,y,z)
  pose_vec_to_matrix / quaternion_to_rotation_matrix /root/reference/hislam2/util/utils.py:676-700     (t, quat x,y,z,w)
  Rotation.from_matrix(R).as_quat()                 /root/reference/hislam2/track_frontend.py:238-239 (scipy, double)
Everything that touches per-pixel data is a HIP kernel (ops.align_view, ops.logdepth_sum, ops.overlap_*).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def pose_encoding_to_camera(enc: np.ndarray) -> np.ndarray:
    """enc [B,7] = (t, quat wxyz) -> c2w [B,4,4] fp32.  No normalisation of the quaternion beyond 2/|q|^2."""
    enc = np.asarray(enc, np.float32).reshape(-1, 7)
    r, i, j, k = enc[:, 3], enc[:, 4], enc[:, 5], enc[:, 6]
    two_s = np.float32(2.0) / (enc[:, 3:7] * enc[:, 3:7]).sum(-1)
    o = np.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                  two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                  two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1).astype(np.float32)
    T = np.tile(np.eye(4, dtype=np.float32), (len(enc), 1, 1))
    T[:, :3, :3] = o.reshape(-1, 3, 3)
    T[:, :3, 3] = enc[:, :3]
    return T


def pose_vec_to_matrix(pose: np.ndarray) -> np.ndarray:
    """pose [B,7] = (t, quat xyzw) -> c2w [B,4,4] fp32 (quaternion normalised first)."""
    pose = np.asarray(pose, np.float32).reshape(-1, 7)
    q = pose[:, 3:] / np.linalg.norm(pose[:, 3:], axis=-1, keepdims=True).astype(np.float32)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                  2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                  2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y], -1).astype(np.float32)
    T = np.tile(np.eye(4, dtype=np.float32), (len(pose), 1, 1))
    T[:, :3, :3] = R.reshape(-1, 3, 3)
    T[:, :3, 3] = pose[:, :3]
    return T


def matrix_to_pose_vec(T: np.ndarray) -> np.ndarray:
    """c2w [4,4] -> (t, quat xyzw) fp32 through scipy's Rotation (as the reference does)."""
    q = Rotation.from_matrix(np.asarray(T[:3, :3], np.float64)).as_quat()
    return np.concatenate([np.asarray(T[:3, 3], np.float32), q.astype(np.float32)])


def matrices_to_pose_vecs(T: np.ndarray) -> np.ndarray:
    """c2w [B,4,4] -> [B,7] (t, quat xyzw) fp32: one vectorised scipy call for a whole window (same per-matrix math)."""
    T = np.asarray(T)
    q = Rotation.from_matrix(np.asarray(T[:, :3, :3], np.float64)).as_quat()
    return np.concatenate([np.asarray(T[:, :3, 3], np.float32), q.astype(np.float32)], axis=1)


def inv4(T: np.ndarray) -> np.ndarray:
    return np.linalg.inv(np.asarray(T, np.float32)).astype(np.float32)


def w2c_rows(c2w: np.ndarray) -> np.ndarray:
    """c2w [B,4,4] -> fp32 [B,12] rows of inverse(c2w)[:3,:4] (float64 inverse, rounded once)."""
    inv = np.linalg.inv(np.asarray(c2w, np.float64).reshape(-1, 4, 4))
    return np.ascontiguousarray(inv[:, :3, :].reshape(-1, 12), dtype=np.float32)


def chain_pose(first_w2c, pose, align_R=None, align_t=None, align_s=None):
    """track_frontend.py:202-234: pose <- first_w2c @ pose, then (optionally) the Sim(3)-style chaining."""
    pose = (first_w2c @ pose).astype(np.float32)
    if align_R is None:
        return pose
    out = np.eye(4, dtype=np.float32)
    out[:3, :3] = align_R @ pose[:3, :3]
    out[:3, 3] = align_R @ (np.float32(align_s) * pose[:3, 3]) + align_t
    return out
