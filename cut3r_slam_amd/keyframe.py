"""Pre-allocated keyframe store, resident in HBM.

Same fields / indexing semantics as the reference `KeyFrame` (/root/reference/hislam2/keyframe.py:5-107), but every
per-pixel buffer lives on the GPU (the reference keeps image/pose/submap_ds/conf_ds/depth on the CPU and re-uploads
all previous pointmaps for every keyframe, track_frontend.py:248-258).  The 7-float poses stay on the host (they are
produced by host 4x4 math) with a device mirror of the 3x4 world->camera rows for the overlap kernels.
"""
from __future__ import annotations

import numpy as np
import torch

from . import geom_host as gh


class KeyFrame:
    def __init__(self, config, image_size, buffer, downsample_ratio, device="cuda:0", feat_dim=1024, patch=16):
        self.ht = ht = int(image_size[0])
        self.wd = wd = int(image_size[1])
        self.buffer = int(buffer)
        self.is_initialized = False
        self.config = config
        self.downsample_ratio = ds = int(downsample_ratio)
        self.device = torch.device(device)
        self._counter = 0
        dev = self.device
        self.tstamp = torch.zeros(buffer, dtype=torch.float)                              # host (a device scalar write would block on the stream)
        self.image = torch.zeros(buffer, 3, ht, wd, device=dev, dtype=torch.uint8)
        self.intrinsic = torch.zeros(buffer, 4, dtype=torch.float)                       # host
        self.pose = torch.zeros(buffer, 7, dtype=torch.float)                            # host, c2w (t, q_xyzw)
        self.pose[:] = torch.as_tensor([0, 0, 0, 0, 0, 0, 1], dtype=torch.float)
        self.w2c = torch.zeros(buffer, 12, device=dev, dtype=torch.float)                # device mirror of inverse(pose)
        nsub = buffer // 5 + 1
        self.submap_ds = torch.ones(nsub, 6, ht // ds, wd // ds, 3, device=dev, dtype=torch.float)
        self.conf_ds = torch.zeros(nsub, 6, ht // ds, wd // ds, device=dev, dtype=torch.float)
        self.depth = torch.ones(buffer, ht, wd, device=dev, dtype=torch.float)
        n = (ht // patch) * (wd // patch)
        self.featI = torch.zeros(buffer, n, feat_dim, dtype=torch.float, device=dev)
        self.pos = torch.zeros(buffer, n, 2, dtype=torch.int64, device=dev)
        self.feat_valid = [False] * buffer            # host flags: featI[i] holds the encoder features of keyframe i

    # the reference exposes an mp.Value; single-process here, same `.counter.value` spelling
    class _Counter:
        def __init__(self, kf):
            self._kf = kf

        @property
        def value(self):
            return self._kf._counter

        @value.setter
        def value(self, v):
            self._kf._counter = int(v)

    @property
    def counter(self):
        return KeyFrame._Counter(self)

    def set_pose(self, index: int, pose7) -> None:
        """store c2w (t, q_xyzw) and refresh the device world->camera mirror used by the overlap kernels"""
        p = torch.as_tensor(np.asarray(pose7, np.float32))
        self.pose[index] = p
        rows = gh.w2c_rows(gh.pose_vec_to_matrix(p.numpy()[None]))
        self.w2c[index].copy_(torch.from_numpy(rows[0]), non_blocking=True)

    def set_poses(self, start: int, poses7, upload: bool = True) -> np.ndarray:
        """set_pose for the consecutive keyframes start..; returns their world->camera rows [n,12].  upload=False leaves
        the device mirror to the caller (ops.window_update writes the rows from kernel arguments: no copy operation)."""
        p = np.asarray(poses7, np.float32).reshape(-1, 7)
        n = p.shape[0]
        self.pose[start:start + n] = torch.from_numpy(p)
        rows = gh.w2c_rows(gh.pose_vec_to_matrix(p))
        if upload:
            self.w2c[start:start + n].copy_(torch.from_numpy(rows), non_blocking=True)
        return rows

    def pointmap_slot(self, kf: int, sub_num: int, t0: int):
        """(submap, slot) holding the current estimate of keyframe `kf` as seen from window `sub_num` starting at t0
        (track_frontend.py:251-255: earlier submaps contribute slots 0..4, the running window its own slots)."""
        if kf >= t0:
            return sub_num, kf - t0
        return kf // 5, kf % 5

    def append(self, tstamp, image, pose, _unused, depth, normal, intrinsics, feat=None, pos=None):
        """keyframe.py:43-77,105-107"""
        i = self._counter
        if i >= self.buffer:
            raise IndexError(f"keyframe buffer overflow ({self.buffer}); raise --buffer")
        self._counter = i + 1
        self.tstamp[i] = float(tstamp)
        self.image[i].copy_(image.to(self.device, non_blocking=True))
        if pose is not None:
            self.set_pose(i, pose)
        if depth is not None:
            self.depth[i].copy_(torch.as_tensor(depth).to(self.device))
        if intrinsics is not None:
            self.intrinsic[i] = torch.as_tensor(intrinsics, dtype=torch.float).reshape(-1)[:4]
        else:
            self.intrinsic[i] = self.intrinsic[0].clone()
        if feat is not None:
            self.featI[i].copy_(feat)
            self.feat_valid[i] = True
        if pos is not None:
            self.pos[i].copy_(pos.reshape(self.pos[i].shape))
