"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's window chaining and covisibility bookkeeping, used
to check the product's HIP-backed tracker on identical network outputs.

  OracleOverlapBackend   counts from oracle/oracle_geom.c (pinned to the reference's cal_overlap_* outputs)
  track_window           /root/reference/hislam2/track_frontend.py:166-262 in torch-CPU fp32, line by line
  predict                /root/reference/hislam2/track_frontend.py:102-162 (2-view relocalisation of a non-keyframe)
  RefGraph               /root/reference/hislam2/factor_graph.py:29-39,59-81,109-117,148-197 (ordered edge lists)
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from . import geom as G


class OracleOverlapBackend:
    """Drop-in for cut3r_slam_amd.factor_graph.HipOverlapBackend, on the CPU (tests only)."""

    def fwd(self, pointmap, w2c_rows, K4, W, H):
        if hasattr(pointmap, "P12"):           # AlignedPoints: materialise P*(s*pts) with the oracle's align
            pts = pointmap.pts.detach().cpu().numpy()
            ones = np.full(pts.shape[:2], 2.0, np.float32)
            pm, _, _ = G.align_view(pts, ones, np.asarray(pointmap.P12, np.float32), pointmap.s, 1)
        else:
            pm = pointmap.detach().cpu().numpy()
        return torch.from_numpy(G.overlap_fwd(pm, w2c_rows.detach().cpu().numpy(), K4, W, H))

    def bwd(self, pointmaps, w2c_row, K4, W, H):
        if hasattr(pointmaps, "store"):        # SubmapStore
            st = pointmaps.store.detach().cpu().numpy()
            pms = np.stack([st[j // 5, j % 5] for j in range(pointmaps.n)])
        else:
            pms = pointmaps.detach().cpu().numpy()
        return torch.from_numpy(G.overlap_bwd(pms, w2c_row.detach().cpu().numpy(), K4, W, H))


def quaternion_to_matrix_wxyz(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def pose_encoding_to_camera(enc):
    R = quaternion_to_matrix_wxyz(enc[:, 3:7])
    T = torch.eye(4)[None].repeat(len(R), 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = enc[:, :3]
    return T


def pose_vec_to_matrix(pose):
    q = pose[:, 3:] / pose[:, 3:].norm(p=2, dim=-1, keepdim=True)
    x, y, z, w = q.unbind(-1)
    R = torch.stack([1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                     2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                     2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y], -1).reshape(-1, 3, 3)
    T = torch.eye(4).repeat(pose.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = pose[:, :3]
    return T


def predict(pts3ds_self, conf_self, pose_enc, kf_pose, kf_depth, ds=2):
    """/root/reference/hislam2/track_frontend.py:102-162 on the outputs of the 2-view inference [keyframe, new frame]
    (torch CPU): returns (new_pose [7] c2w (t, q_xyzw), new_depth [H,W], new_pointmap [h,w,3], new_conf [h,w]).
    Quirk kept: the returned confidence is the raw conf_self of view 0 (the keyframe), downsampled (:114,157)."""
    poses = pose_encoding_to_camera(pose_enc)
    depths = pts3ds_self[..., 2]
    conf = conf_self[0]
    log_scale = (torch.log(kf_depth) - torch.log(depths[0])).mean()
    align_s = torch.exp(log_scale)
    prev_c2w = pose_vec_to_matrix(kf_pose.unsqueeze(0))[0]
    align_R, align_t = prev_c2w[:3, :3], prev_c2w[:3, 3]
    pose = torch.inverse(poses[0]) @ poses[1]
    pa = torch.eye(4)
    pa[:3, :3] = align_R @ pose[:3, :3]
    pa[:3, 3] = align_R @ (align_s * pose[:3, 3]) + align_t
    pointmap = torch.einsum("ij,hwj->hwi", pa[:3, :3], align_s * pts3ds_self[1]) + pa[:3, 3]
    quat = torch.from_numpy(Rotation.from_matrix(pa[:3, :3].numpy()).as_quat())
    return torch.cat([pa[:3, 3], quat], dim=-1), align_s * depths[1], pointmap[::ds, ::ds], conf[::ds, ::ds]


def track_window(state, t0, t1, pts3ds_self, conf_self, pose_enc, init, ds=2):
    """state: dict with 'pose' [buf,7], 'depth' [buf,H,W], 'submap_ds' [S,6,h,w,3], 'conf_ds' [S,6,h,w] (torch CPU).
    Mutates state exactly as the reference's loop body (without the graph calls) and returns per-view
    (pose 4x4, scale) so the caller can drive a graph with them."""
    poses = pose_encoding_to_camera(pose_enc)
    depths = pts3ds_self[..., 2]
    first_w2c = torch.inverse(poses[0])
    sub_num = t0 // 5
    out = []
    align_s = align_R = align_t = None
    for i in range(t0, t1):
        pts = pts3ds_self[i - t0]
        conf = 1 - 1 / conf_self[i - t0]
        depth = depths[i - t0]
        pose = poses[i - t0]
        if init:
            pose = first_w2c @ pose
            pointmap = torch.einsum("ij,hwj->hwi", pose[:3, :3], pts) + pose[:3, 3]
            s = 1.0
        else:
            if i == t0:
                log_scale = (torch.log(state["depth"][i]) - torch.log(depth)).mean()
                align_s = torch.exp(log_scale)
                prev_c2w = pose_vec_to_matrix(state["pose"][i].unsqueeze(0))[0]
                align_R, align_t = prev_c2w[:3, :3], prev_c2w[:3, 3]
            pose = first_w2c @ pose
            R, T = pose[:3, :3], pose[:3, 3]
            pa = torch.eye(4)
            pa[:3, :3] = align_R @ R
            pa[:3, 3] = align_R @ (align_s * T) + align_t
            pointmap = torch.einsum("ij,hwj->hwi", pa[:3, :3], align_s * pts) + pa[:3, 3]
            depth = align_s * depth
            pose = pa
            s = float(align_s)
        quat = torch.from_numpy(Rotation.from_matrix(pose[:3, :3].numpy()).as_quat())
        state["submap_ds"][sub_num, i - t0] = pointmap[::ds, ::ds]
        state["conf_ds"][sub_num, i - t0] = conf[::ds, ::ds]
        state["pose"][i] = torch.cat([pose[:3, 3], quat], dim=-1)
        state["depth"][i] = depth
        out.append((pose.clone(), s, pointmap))
    return out
