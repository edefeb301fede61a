"""Tensor helpers the reference trackers import from the model package (same names, argument meaning and error behaviour):

    pose_encoding_to_camera / quaternion_to_matrix   /root/reference/src/dust3r/utils/camera.py:364-420   (t(3), quat w,x,y,z -> c2w 4x4)
    geotrf / inv                                      /root/reference/src/dust3r/utils/geometry.py:49-124

called by hislam2/track_frontend.py:9,11 and hislam2/track_backend.py:8,9 on the network outputs.  They are thin torch (or numpy)
expressions with no kernel of their own: the product's own trackers do this arithmetic on the host in fp64 (`geom_host.py`) or inside
`cut3r_window_update`; these versions exist so that the reference's tracker files run against this package WITHOUT an edit
(`cut3r_slam_amd/compat/src/dust3r/utils/{camera,geometry}.py` re-export them).  Checked against the reference's own functions on the
CPU through tests/golden/camera.npz (tests/test_compat_imports_cpu.py).
"""
from __future__ import annotations

import numpy as np
import torch


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    """(..., 4) real-part-first quaternions (not necessarily unit) -> (..., 3, 3) rotation matrices"""
    w, x, y, z = torch.unbind(quaternions, -1)
    s = 2.0 / (quaternions * quaternions).sum(-1)
    rows = (1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
            s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
            s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y))
    return torch.stack(rows, -1).reshape(quaternions.shape[:-1] + (3, 3))


def pose_encoding_to_camera(pose_encoding: torch.Tensor, pose_encoding_type: str = "absT_quaR") -> torch.Tensor:
    """[B, >=7] pose encodings (translation, quaternion w,x,y,z) -> [B,4,4] camera-to-world matrices"""
    if pose_encoding_type != "absT_quaR":
        raise ValueError(f"Unknown pose encoding {pose_encoding_type}")
    R = quaternion_to_matrix(pose_encoding[:, 3:7])
    c2w = torch.eye(4, dtype=R.dtype, device=R.device).repeat(len(R), 1, 1)
    c2w[:, :3, :3] = R
    c2w[:, :3, 3] = pose_encoding[:, :3]
    return c2w


def geotrf(Trf, pts, ncol=None, norm=False):
    """Apply the d x d or (d+1) x (d+1) transformation(s) `Trf` to points `pts` (..., d): the reference's general routine.  A batch of
    matrices [B,·,·] with a batch of point MAPS [B,H,W,d] takes the einsum path (affine part + translation, no projective row); every
    other combination multiplies row vectors by the transposed matrix; `norm` projects onto the plane z = norm."""
    assert Trf.ndim >= 2
    if isinstance(Trf, np.ndarray):
        pts = np.asarray(pts)
    elif isinstance(Trf, torch.Tensor):
        pts = torch.as_tensor(pts, dtype=Trf.dtype)
    lead = pts.shape[:-1]
    d = pts.shape[-1]
    ncol = ncol or d
    if isinstance(Trf, torch.Tensor) and isinstance(pts, torch.Tensor) and Trf.ndim == 3 and pts.ndim == 4:
        if Trf.shape[-1] == d:
            pts = torch.einsum("bij, bhwj -> bhwi", Trf, pts)
        elif Trf.shape[-1] == d + 1:
            pts = torch.einsum("bij, bhwj -> bhwi", Trf[:, :d, :d], pts) + Trf[:, None, None, :d, d]
        else:
            raise ValueError(f"bad shape, not ending with 3 or 4, for {pts.shape=}")
    else:
        if Trf.ndim >= 3:
            n = Trf.ndim - 2
            assert Trf.shape[:n] == pts.shape[:n], "batch size does not match"
            Trf = Trf.reshape(-1, Trf.shape[-2], Trf.shape[-1])
            if pts.ndim > Trf.ndim:
                pts = pts.reshape(Trf.shape[0], -1, d)          # [B, H, W, d] -> [B, H*W, d]
            elif pts.ndim == 2:
                pts = pts[:, None, :]                           # [B, d] -> [B, 1, d]
        if d + 1 == Trf.shape[-1]:
            T = Trf.swapaxes(-1, -2)
            pts = pts @ T[..., :-1, :] + T[..., -1:, :]
        elif d == Trf.shape[-1]:
            pts = pts @ Trf.swapaxes(-1, -2)
        else:
            pts = Trf @ pts.T
            if pts.ndim >= 2:
                pts = pts.swapaxes(-1, -2)
    if norm:
        pts = pts / pts[..., -1:]
        if norm != 1:
            pts = pts * norm
    return pts[..., :ncol].reshape(*lead, ncol)


def inv(mat):
    """inverse of a torch or numpy matrix (batch)"""
    if isinstance(mat, torch.Tensor):
        return torch.linalg.inv(mat)
    if isinstance(mat, np.ndarray):
        return np.linalg.inv(mat)
    raise ValueError(f"bad matrix type = {type(mat)}")
