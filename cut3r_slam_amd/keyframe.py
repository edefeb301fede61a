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


class _FeatValid:
    """`feat_valid[i]`: does the feature store hold keyframe i?  Full store: one flag per keyframe; ring: row tags."""

    def __init__(self, buffer, rows):
        self.rows = rows
        self.flags = [False] * buffer if not rows else None
        self.tags = [-1] * rows if rows else None

    def __getitem__(self, i):
        return self.flags[i] if self.flags is not None else self.tags[i % self.rows] == i

    def __setitem__(self, i, v):
        if self.flags is not None:
            self.flags[i] = bool(v)
        elif v:
            self.tags[i % self.rows] = i
        elif self.tags[i % self.rows] == i:
            self.tags[i % self.rows] = -1


class KeyFrame:
    def __init__(self, config, image_size, buffer, downsample_ratio, device="cuda:0", feat_dim=1024, patch=16, feat_buffer=0):
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
        # encoder features: one row per keyframe like the reference (keyframe.py:36) -- or, for throughput drivers that only ever
        # need the features of the windows in flight (no loop closure), a RING of `feat_buffer` rows: keyframe i lives in row
        # i % feat_buffer, rows 0..5 are mirrored behind the end so that any 6-keyframe window is one contiguous slice
        self.feat_rows = int(feat_buffer) if 0 < int(feat_buffer) < buffer else 0
        self.featI = torch.zeros(self.feat_rows + 6 if self.feat_rows else buffer, n, feat_dim, dtype=torch.float, device=dev)
        self.pos = torch.zeros(buffer if not self.feat_rows else 1, n, 2, dtype=torch.int64, device=dev)
        self.feat_valid = _FeatValid(buffer, self.feat_rows)   # host flags: the store holds the encoder features of keyframe i

    def feat_store(self, a: int, b: int, feats) -> None:
        """features of keyframes a..b-1 (one batch of the encoder) into the store"""
        R = self.feat_rows
        if not R:
            self.featI[a:b] = feats
        else:
            if b - a > R:
                raise ValueError("feature ring smaller than one encoder batch")
            i = a
            while i < b:                                   # contiguous pieces up to the wrap
                r = i % R
                m = min(b - i, R - r)
                self.featI[r:r + m] = feats[i - a:i - a + m]
                if r < 6:                                  # mirror of rows 0..5
                    k = min(m, 6 - r)
                    self.featI[R + r:R + r + k] = feats[i - a:i - a + k]
                i += m
        for i in range(a, b):
            self.feat_valid[i] = True

    def feat_slice(self, t0: int, t1: int):
        """features of keyframes t0..t1-1 (at most 6 in ring mode) as one contiguous slice"""
        R = self.feat_rows
        if not R:
            return self.featI[t0:t1]
        if t1 - t0 > 6:
            raise ValueError("feature ring: windows of at most 6 keyframes")
        r = t0 % R
        return self.featI[r:r + (t1 - t0)]

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

    def set_poses_at(self, idx, poses7):
        """set_pose for arbitrary keyframe indices (the mapper's write-back, hi2.py:84): host table + device mirror"""
        p = np.asarray(poses7, np.float32).reshape(-1, 7)
        rows = gh.w2c_rows(gh.pose_vec_to_matrix(p))
        for j, k in enumerate(idx):
            self.pose[int(k)] = torch.from_numpy(p[j])
            self.w2c[int(k)].copy_(torch.from_numpy(rows[j]))

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
        if image is not None:            # (None: a keyframe another rank encodes -- registered, its pixels never read here)
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
            self.feat_store(i, i + 1, feat[None] if feat.dim() == 2 else feat)
        if pos is not None and not self.feat_rows:
            self.pos[i].copy_(pos.reshape(self.pos[i].shape))
