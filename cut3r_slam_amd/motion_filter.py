"""Keyframe selection with the reference's control flow (/root/reference/hislam2/motion_filter.py:70-135):
first / second-last / last frame are always keyframes; otherwise every `skip`-th frame is encoded and becomes a
keyframe when its patch-overlap ratio against the last keyframe drops below `thresh` (overlap mode, kf_every <= 0),
or every `kf_every`-th frame is taken unconditionally (fixed-cadence mode).

The encoder pass is the HIP ViT-L (cut3r_slam_amd.model); the overlap test is one exact-fp32 MFMA kernel
(ops.patch_overlap_count) whose only host traffic is the 4-byte count the decision needs.
"""
from __future__ import annotations

import torch

from . import ops


class MotionFilter:
    def __init__(self, model, keyframes, config, device="cuda:0"):
        self.model = model
        self.keyframes = keyframes
        self.thresh = config["thresh"]
        self.skip = config.get("skip", 1)
        self.init_thresh = config.get("init_thresh", self.thresh)
        self.kf_every = config.get("kf_every", -1)
        self.skip_blur = config.get("skip_blur", False)
        self.device = device
        self._count = torch.zeros(1, dtype=torch.int32, device=device)
        self._ws = None

    def encode(self, image_u8):
        """image_u8 [1,3,H,W] uint8 (host or device) -> (feat [N,C] fp32, pos [1,N,2])"""
        img = image_u8.to(self.device, non_blocking=True)
        feat, pos, _ = self.model.encode_image({"img": img})       # uint8: normalisation fused into the patch loader
        return feat[0], pos

    def overlap_ratio(self, feat0, feat1, threshold=0.7):
        N, C = feat0.shape
        need = 2 * (N - 1) * C + N
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device)
        ops.patch_overlap_count(feat0.contiguous(), feat1.contiguous(), threshold, self._ws, self._count)
        return float(self._count.item()) / float(N - 1)            # matched.mean().item(), utils.py:733-734

    @torch.no_grad()
    def kfFilter(self, tstamp, image, intrinsics=None, pose=None, depth=None, second_last_frame=False, last_frame=False):
        kf = self.keyframes
        compute_overlap = not (self.kf_every > 0)
        if kf.counter.value == 0 or last_frame or second_last_frame:
            feat1, pos1 = (self.encode(image[:1]) if compute_overlap else (None, None))
            kf.append(tstamp, image[0], pose, 1.0, depth, None, intrinsics, feat1, pos1)
            return True
        overlap_ratio, feat1, pos1 = 1.0, None, None
        if compute_overlap and tstamp % self.skip == 0:
            feat0 = kf.featI[kf.counter.value - 1]
            feat1, pos1 = self.encode(image[:1])
            overlap_ratio = self.overlap_ratio(feat0, feat1)
        elif not compute_overlap and tstamp % self.kf_every == 0:
            # fixed cadence: the decision does not depend on the features, so the encoder pass is deferred to the
            # tracking window, which encodes its new keyframes as ONE batch and stores them in keyframes.featI
            feat1, pos1 = None, None
        if (compute_overlap and overlap_ratio < self.thresh) or (not compute_overlap and tstamp % self.kf_every == 0):
            kf.append(tstamp, image[0], pose, None, depth, None, intrinsics, feat1, pos1)
            return True
        return False
