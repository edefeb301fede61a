"""Keyframe selection with the reference's control flow (/root/reference/hislam2/motion_filter.py:70-135):
first / second-last / last frame are always keyframes; otherwise every `skip`-th frame is encoded and becomes a
keyframe when its patch-overlap ratio against the last keyframe drops below `thresh` (overlap mode, kf_every <= 0),
or every `kf_every`-th frame is taken unconditionally (fixed-cadence mode).

The encoder pass is the HIP ViT-L (cut3r_slam_amd.model); the overlap test is one exact-fp32 MFMA kernel
(ops.patch_overlap_count) whose only host traffic is the 4-byte count the decision needs.

Buffered streams (`prefetch`): the encoder is batch-invariant bit for bit (tests/test_model_gpu.py), so the next k tested
frames can go through it as ONE batch, and their decisions -- a sequential scan, each frame against the last keyframe so
far -- are taken on the device by ops.patch_overlap_chain with the same arithmetic; the host reads k decisions back in one
copy.  `kfFilter` then consumes the cached features and decisions: same keyframes, same features, no B=1 encoder launches
and no per-frame `.item()`.
"""
from __future__ import annotations

import numpy as np
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
        self._chain_ws = None
        self._ahead = {}                 # tstamp -> (features [N,C], decision, count) from the last prefetch
        self._pos_grid = None
        self._ahead_base = None          # keyframe counter the cached decisions were taken against
        self.last_ratio = None
        self._inflight = None            # event of a prefetch_launch whose result has not been collected
        self.stats = {"encoded": 0, "prefetched": 0, "cache_hits": 0}

    def encode(self, image_u8):
        """image_u8 [1,3,H,W] uint8 (host or device) -> (feat [N,C] fp32, pos [1,N,2])"""
        if self._inflight is not None:       # a look-ahead pass on its side stream uses the encoder's static buffers: never beside it
            self._inflight.synchronize()
            self._inflight = None
        img = image_u8.to(self.device, non_blocking=True)
        feat, pos, _ = self.model.encode_image({"img": img})       # uint8: normalisation fused into the patch loader
        self.stats["encoded"] += 1
        return feat[0], pos

    @staticmethod
    def ratio_of(count: int, n_rows: int) -> float:
        """`matched.mean().item()` (hislam2/util/utils.py:733-734): an fp32 mean read back as a Python float"""
        return float(np.float32(count) / np.float32(n_rows))

    def overlap_ratio(self, feat0, feat1, threshold=0.7):
        N, C = feat0.shape
        need = 2 * (N - 1) * C + N
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device)
        ops.patch_overlap_count(feat0.contiguous(), feat1.contiguous(), threshold, self._ws, self._count)
        return self.ratio_of(int(self._count.item()), N - 1)

    # ------------------------------------------------------------------ look-ahead over a buffered stream
    @torch.no_grad()
    def prefetch(self, images_u8, tstamps, forced=None):
        """Encode the tested frames `images_u8` [B,3,H,W] (time stamps `tstamps`, increasing, every one a multiple of `skip`)
        as one batch and take their keyframe decisions on the device.  forced[i] marks frames the filter always keeps
        (first / second-last / last frame).  Valid until a keyframe is appended that this scan did not foresee."""
        if self.kf_every > 0 or len(tstamps) == 0:
            return
        kf = self.keyframes
        B = len(tstamps)
        if self._inflight is not None:
            self._inflight.synchronize()
            self._inflight = None
        imgs = images_u8.to(self.device, non_blocking=True)
        feats = self.model.encode_batch(imgs)                                             # [B,N,C] fp32, one batched pass
        self.stats["prefetched"] += B
        N, C = feats.shape[1:]
        P = self.model.cfg.patch_size
        nh, nw = imgs.shape[2] // P, imgs.shape[3] // P
        if self._pos_grid is None or self._pos_grid.shape[1] != nh * nw:                  # PositionGetter (croco/models/blocks.py:323-335)
            y, x = torch.meshgrid(torch.arange(nh, device=self.device), torch.arange(nw, device=self.device), indexing="ij")
            self._pos_grid = torch.stack([y.reshape(-1), x.reshape(-1)], -1)[None]
        forced = [bool(f) for f in forced] if forced is not None else [False] * B
        if kf.counter.value == 0:
            forced[0] = True
            feat_last = feats[0]
        else:
            feat_last = kf.feat_slice(kf.counter.value - 1, kf.counter.value)[0]
        need = (B + 1) * (N - 1) * C + N
        if self._chain_ws is None or self._chain_ws[0].numel() < need or self._chain_ws[1].numel() < 2 * B + 1:
            self._chain_ws = (torch.empty(need, device=self.device), torch.zeros(2 * B + 1, dtype=torch.int32, device=self.device))
        ws, ints = self._chain_ws
        ops.patch_overlap_chain(feat_last.contiguous(), feats, 0.7, float(self.thresh), forced, ws, ints[2 * B:2 * B + 1], ints[:B], ints[B:2 * B])
        host = ints[:2 * B].cpu().numpy()                                                 # THE device round trip of the batch
        self._ahead = {int(t): (feats[i], bool(host[B + i]), int(host[i]), forced[i]) for i, t in enumerate(tstamps)}
        self._ahead_base = kf.counter.value

    @torch.no_grad()
    def prefetch_launch(self, images_u8, tstamps, forced=None, feat_last=None, stream=None):
        """`prefetch` in two halves, the first one asynchronous: encode the tested frames and run the decision chain on `stream` (a side stream:
        the pass then runs BESIDE whatever the caller issues next on the current stream -- the previous chunk's tracking windows), with the
        decisions copied to pinned host memory behind an event.  `feat_last`: the features the chain starts from when they are not the
        store's last keyframe (the caller knows the last keyframe of the chunk still being consumed).  Returns the handle for
        `prefetch_collect`; nothing of the filter's state changes until then."""
        if self.kf_every > 0 or len(tstamps) == 0:
            return None
        kf = self.keyframes
        B = len(tstamps)
        cur = torch.cuda.current_stream()
        st = stream if stream is not None else cur
        if st is not cur:
            st.wait_stream(cur)                      # frames, keyframe store: everything issued so far
        with torch.cuda.stream(st):
            imgs = images_u8.to(self.device, non_blocking=True)
            feats = self.model.encode_batch(imgs)                                         # [B,N,C] fp32, a fresh tensor
            N, C = feats.shape[1:]
            P = self.model.cfg.patch_size
            nh, nw = imgs.shape[2] // P, imgs.shape[3] // P
            if self._pos_grid is None or self._pos_grid.shape[1] != nh * nw:
                y, x = torch.meshgrid(torch.arange(nh, device=self.device), torch.arange(nw, device=self.device), indexing="ij")
                self._pos_grid = torch.stack([y.reshape(-1), x.reshape(-1)], -1)[None]
            forced = [bool(f) for f in forced] if forced is not None else [False] * B
            if feat_last is None:
                if kf.counter.value == 0:
                    forced[0] = True
                    feat_last = feats[0]
                else:
                    feat_last = kf.feat_slice(kf.counter.value - 1, kf.counter.value)[0]
            need = (B + 1) * (N - 1) * C + N
            if self._chain_ws is None or self._chain_ws[0].numel() < need or self._chain_ws[1].numel() < 2 * B + 1:
                self._chain_ws = (torch.empty(need, device=self.device), torch.zeros(2 * B + 1, dtype=torch.int32, device=self.device))
            ws, ints = self._chain_ws
            ops.patch_overlap_chain(feat_last.contiguous(), feats, 0.7, float(self.thresh), forced, ws, ints[2 * B:2 * B + 1], ints[:B], ints[B:2 * B])
            host = torch.empty(2 * B, dtype=torch.int32).pin_memory()
            host.copy_(ints[:2 * B], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        self.stats["prefetched"] += B
        self._inflight = ev
        return {"feats": feats, "host": host, "event": ev, "tstamps": [int(t) for t in tstamps], "forced": forced}

    def prefetch_collect(self, handle, base=None):
        """second half of `prefetch_launch`: wait for the decisions (THE device round trip of the batch) and install them.  `base`: the keyframe
        counter the decisions are valid against (default: now).  Returns (number of frames the scan keeps, features of the last kept one | None)."""
        if handle is None:
            return 0, None
        handle["event"].synchronize()
        if self._inflight is handle["event"]:
            self._inflight = None
        host = handle["host"].numpy()
        feats, ts, forced = handle["feats"], handle["tstamps"], handle["forced"]
        B = len(ts)
        self._ahead = {t: (feats[i], bool(host[B + i]), int(host[i]), forced[i]) for i, t in enumerate(ts)}
        self._ahead_base = self.keyframes.counter.value if base is None else int(base)
        kept = [i for i in range(B) if bool(host[B + i]) or forced[i]]
        return len(kept), (feats[kept[-1]] if kept else None)

    def _cached(self, tstamp):
        """cached (features, decision, count) of a tested frame, if the scan that produced them still holds"""
        ent = self._ahead.get(int(tstamp))
        if ent is None:
            return None
        if self.keyframes.counter.value != self._ahead_base:          # a keyframe the scan did not know about: rescan from here
            self._ahead = {}
            return None
        return ent

    def _consume(self, tstamp, took):
        self._ahead.pop(int(tstamp), None)
        if took:
            self._ahead_base = self.keyframes.counter.value
        self.stats["cache_hits"] += 1

    @torch.no_grad()
    def kfFilter(self, tstamp, image, intrinsics=None, pose=None, depth=None, second_last_frame=False, last_frame=False):
        kf = self.keyframes
        compute_overlap = not (self.kf_every > 0)
        if kf.counter.value == 0 or last_frame or second_last_frame:
            ent = self._cached(tstamp) if compute_overlap else None
            if ent is not None and ent[3]:                       # the look-ahead knew this frame is always kept
                feat1, pos1 = ent[0], self._pos_grid
                kf.append(tstamp, image[0], pose, 1.0, depth, None, intrinsics, feat1, pos1)
                self._consume(tstamp, True)
                return True
            self._ahead = {}
            feat1, pos1 = (self.encode(image[:1]) if compute_overlap else (None, None))
            kf.append(tstamp, image[0], pose, 1.0, depth, None, intrinsics, feat1, pos1)
            return True
        overlap_ratio, feat1, pos1 = 1.0, None, None
        if compute_overlap and tstamp % self.skip == 0:
            ent = self._cached(tstamp)
            if ent is not None and not ent[3]:
                feat1, took, count, _ = ent
                self.last_ratio = self.ratio_of(count, feat1.shape[0] - 1)
                if took:
                    kf.append(tstamp, image[0], pose, None, depth, None, intrinsics, feat1, self._pos_grid)
                self._consume(tstamp, took)
                return took
            feat0 = kf.feat_slice(kf.counter.value - 1, kf.counter.value)[0]
            feat1, pos1 = self.encode(image[:1])
            overlap_ratio = self.last_ratio = self.overlap_ratio(feat0, feat1)
        elif not compute_overlap and tstamp % self.kf_every == 0:
            # fixed cadence: the decision does not depend on the features, so the encoder pass is deferred to the
            # tracking window, which encodes its new keyframes as ONE batch and stores them in keyframes.featI
            feat1, pos1 = None, None
        if (compute_overlap and overlap_ratio < self.thresh) or (not compute_overlap and tstamp % self.kf_every == 0):
            kf.append(tstamp, image[0], pose, None, depth, None, intrinsics, feat1, pos1)
            return True
        return False
