"""Tiny host-side (numpy) geometry of the tracking loop: 7-float poses <-> 4x4 matrices, chaining.

These are O(1)-sized computations that the reference also does on the host after `.cpu()`:
  pose_encoding_to_camera / quaternion_to_matrix   /root/reference/src/dust3r/utils/camera.py:364-420  (quat w,x,y,z)
  pose_vec_to_matrix / quaternion_to_rotation_matrix /root/reference/hislam2/util/utils.py:676-700     (t, quat x,y,z,w)
  Rotation.from_matrix(R).as_quat()                 /root/reference/hislam2/track_frontend.py:238-239 (scipy, double)
Everything that touches per-pixel data is a HIP kernel (ops.align_view, ops.logdepth_sum, ops.overlap_*).
"""
from __future__ import annotations

import math

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


def compose_chain(A: np.ndarray, s, rel: np.ndarray) -> np.ndarray:
    """track_frontend.py:224-231 for a stack: A [...,4,4] (c2w of the shared keyframe), s [...] (window scale), rel [...,4,4]
    (first_w2c @ pose) -> R = A_R rel_R, t = A_R (s rel_t) + A_t"""
    A, rel = np.asarray(A, np.float32), np.asarray(rel, np.float32)
    out = np.zeros(rel.shape, np.float32)
    out[..., 3, 3] = 1.0
    AR = A[..., :3, :3]
    out[..., :3, :3] = AR @ rel[..., :3, :3]
    st = (np.asarray(s, np.float32)[..., None] * rel[..., :3, 3])
    out[..., :3, 3] = (AR @ st[..., None])[..., 0] + A[..., :3, 3]
    return out


def chain_windows(encs: np.ndarray, scales_fn, pose_t0: np.ndarray, reset=None):
    """Host scan over consecutive windows (multi-GPU replay): encs [n,V,7] raw pose encodings (t, q_wxyz); scales_fn(k) -> the
    window's fp32 scale (called in order); pose_t0 [7] the stored pose (t, q_xyzw) of the first window's first keyframe.
    Sequential work per window: ONE pose composition and ONE matrix->quaternion conversion (the shared keyframe); everything else
    is batched over all n*V views.  reset[k] true: window k starts a new sequence (TrackFrontend.sequence_windows): it is chained to
    the identity pose (scales_fn returns 1 for it).  Returns (chained [n,V,4,4], scales [n] fp32, pose_vecs [n,V,7], w2c_rows [n,V,12])."""
    encs = np.asarray(encs, np.float32)
    n, V = encs.shape[:2]
    poses = pose_encoding_to_camera(encs.reshape(-1, 7)).reshape(n, V, 4, 4)
    first_w2c = np.linalg.inv(poses[:, 0]).astype(np.float32)
    rel = (first_w2c[:, None] @ poses).astype(np.float32)
    A_all = np.zeros((n, 4, 4), np.float32)
    s_all = np.zeros(n, np.float32)
    A_all[:, 3, 3] = 1.0
    # the sequential part in plain Python floats (numpy / scipy calls on single 4x4 matrices cost 5-100 us each: the per-rank
    # host time of this loop is what bounds the replay at 8 GPUs x 8 windows); values are rounded to fp32 where the stored
    # 7-float pose is (t, q in fp32)
    prev = [float(np.float32(v)) for v in np.asarray(pose_t0, np.float64).reshape(7)]
    relv = rel[:, V - 1].astype(np.float64).tolist()
    for k in range(n):
        if reset is not None and reset[k]:
            prev = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
        x, y, z, w = prev[3:]
        nq = math.sqrt(x * x + y * y + z * z + w * w)
        x, y, z, w = x / nq, y / nq, z / nq, w / nq
        R = [[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
             [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
             [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]]
        R = [[float(np.float32(v)) for v in row] for row in R]
        t = prev[:3]
        sk = float(np.float32(scales_fn(k)))
        for a in range(3):
            A_all[k, a, :3] = R[a]
            A_all[k, a, 3] = t[a]
        s_all[k] = sk
        # the shared keyframe's next pose: A o (s, rel_k[V-1]) -> (t, q_xyzw), quaternion by the largest-diagonal rule
        # (the formula scipy.spatial.transform.Rotation.from_matrix applies)
        rm = relv[k]
        M = [[sum(R[a][c] * rm[c][b] for c in range(3)) for b in range(3)] for a in range(3)]
        tn = [sum(R[a][c] * (sk * rm[c][3]) for c in range(3)) + t[a] for a in range(3)]
        M = [[float(np.float32(v)) for v in row] for row in M]
        tr = M[0][0] + M[1][1] + M[2][2]
        dec = [M[0][0], M[1][1], M[2][2], tr]
        c = max(range(4), key=lambda i_: dec[i_])
        q = [0.0, 0.0, 0.0, 0.0]
        if c != 3:
            i_, j_, k_ = c, (c + 1) % 3, (c + 2) % 3
            q[i_] = 1 - tr + 2 * M[i_][i_]
            q[j_] = M[j_][i_] + M[i_][j_]
            q[k_] = M[k_][i_] + M[i_][k_]
            q[3] = M[k_][j_] - M[j_][k_]
        else:
            q = [M[2][1] - M[1][2], M[0][2] - M[2][0], M[1][0] - M[0][1], 1 + tr]
        nq = math.sqrt(sum(v * v for v in q))
        prev = [float(np.float32(v)) for v in tn] + [float(np.float32(v / nq)) for v in q]
    chained = compose_chain(A_all[:, None], s_all[:, None], rel)
    vecs = matrices_to_pose_vecs(chained.reshape(-1, 4, 4))
    rows = w2c_rows(pose_vec_to_matrix(vecs))
    return chained, s_all, vecs.reshape(n, V, 7), rows.reshape(n, V, 12)
