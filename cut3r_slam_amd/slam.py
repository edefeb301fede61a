"""Tracking-only SLAM orchestrator with the reference's per-frame entry point
(/root/reference/hislam2/hi2.py:17-54 `Hi2.__init__`, :101-133 `Hi2.run`, demo_s.py:97-113 `save_trajectory`).

Scope: keyframe filter -> window tracker -> covisibility graph (-> loop closure when enabled).
The Gaussian-splatting mapper of the reference (`GSBackEnd`, hi2.py:47,82) is out of scope: BASELINE configs 2-4 run
"tracking only, GS backend off".
"""
from __future__ import annotations

import numpy as np
import torch

from .factor_graph import FactorGraph
from .keyframe import KeyFrame
from .motion_filter import MotionFilter
from .track_frontend import TrackFrontend
from .track_backend import TrackBackend

DEFAULT_CONFIG = {
    "Tracking": {
        "motion_filter": {"thresh": 0.9, "skip": 5, "skip_blur": False, "kf_every": -1, "init_thresh": 4.0},
        "frontend": {"keyframe_thresh": 4.0, "frontend_thresh": 16.0, "frontend_window": 25, "frontend_radius": 2,
                     "frontend_nms": 1, "mono_depth_alpha": 0.01, "iteration": 0},
    }
}


class Cut3rSlam:
    def __init__(self, model, config=None, image_size=(384, 512), buffer=512, device="cuda:0", feat_buffer=0):
        self.model = model
        self.config = config or DEFAULT_CONFIG
        self.device = device
        self.verbose = False
        self.output_dir = None
        self.use_gt = False
        self.downsample_ratio = 2
        self.keyframes = KeyFrame(self.config, image_size, buffer, self.downsample_ratio, device,
                                  feat_dim=model.cfg.enc_embed_dim, patch=model.cfg.patch_size, feat_buffer=feat_buffer)
        self.graph = FactorGraph(self.keyframes, device=device, max_factors=48)
        self.filterx = MotionFilter(model, self.keyframes, self.config["Tracking"]["motion_filter"], device)
        self.tracker = TrackFrontend(self, self.keyframes, self.config["Tracking"]["frontend"], device)
        self.backend = TrackBackend(self, self.keyframes, self.config["Tracking"]["frontend"], device)
        self.do_lc = self.config["Tracking"]["frontend"].get("iteration", 0) > 0
        self.freeze_counter = 0
        # hi2.py:103 keeps EVERY full-resolution frame (`self.images[tstamp] = image`) for terminate(); here that is opt-in
        # (frames stay on the device, 0.6 MB each at 384x512): demo.py switches it on when --add_kf semantics are wanted
        self.keep_images = False
        self._ahead_stream = None              # run_buffered(pipeline=True): the look-ahead encoder pass of the next chunk
        self.images = {}
        # dist.ShardedTracker registers keyframes ahead of the tracker (encoder look-ahead): it sets this, and the trajectory writers
        # then stop at tracker.t1 instead of the reference's counter - 1 (demo_s.py:97-100)
        self.tracked_only = False
        # hi2.py:47-48: the Gaussian mapper (cut3r_slam_amd.gs_mapper.GSMapper); None = tracking only (BASELINE configs 1-4)
        self.mapper = None
        self.gs_iter_num = self.config.get("Mapping", {}).get("itr_num", 100)
        # hi2.py:84 reads `self.keyframes.depth[updated_idx][mask] = depth[mask]` with a LIST index: an advanced-index copy, so the
        # reference never changes keyframes.depth and later windows chain on the tracker's own depths.  Default = that effective
        # behaviour; True writes the mapper's scale-corrected depths back (what the line looks like it wants) -- a declared deviation
        self.gs_depth_writeback = False

    def call_gs(self, viz_idx, submap_idx, iterations, intrinsics):
        """hi2.py:56-91: hand the window's keyframes to the mapper and take back its refined poses and pointmaps (the chain of later
        windows then starts from them; depths only with `gs_depth_writeback`, see __init__).  Declared deviation: the images are the
        stored tracking-resolution keyframes with the tracking intrinsics (the reference hands over its full-resolution
        `self.images[t]` with the full-resolution calibration, hi2.py:60-63)."""
        kf = self.keyframes
        viz = list(viz_idx)
        n = len(viz)
        data = {"viz_idx": viz, "submap_idx": submap_idx, "tstamp": kf.tstamp[viz], "poses": kf.pose[viz], "images": kf.image[viz],
                "pointmaps": kf.submap_ds[submap_idx][:n], "confs": kf.conf_ds[submap_idx][:n], "depths": kf.depth[viz],
                "intrinsics": torch.as_tensor(intrinsics).reshape(-1)[:4]}
        with torch.enable_grad():
            updated, idx = self.mapper.run(data, iterations)
        kf.set_poses_at(idx, updated["poses"].float().cpu().numpy())
        depth = updated["depths"]
        ds = self.downsample_ratio
        for j, k in enumerate(idx):
            if getattr(self, "gs_depth_writeback", False):
                ok = depth[j] > 0
                kf.depth[k][ok] = depth[j][ok]
            kf.submap_ds[k // 5, k % 5] = updated["pointmaps"][j, ::ds, ::ds]
        kf.submap_ds[:submap_idx + 1, -1] = kf.submap_ds[1:submap_idx + 2, 0]
        return idx

    @torch.no_grad()
    def run(self, tstamp, image, intrinsics, image_ds, intrinsics_ds, second_last_frame=False, last_frame=False):
        """hi2.py:101-133 (the GS mapper takes part when `self.mapper` is set): image_ds [1,3,H,W] uint8 at tracking resolution."""
        if self.keep_images:
            self.images[int(tstamp)] = image
        self.filterx.kfFilter(tstamp, image_ds, intrinsics=intrinsics_ds, second_last_frame=second_last_frame,
                              last_frame=last_frame)
        lc_did = False
        hook, done = None, []
        if self.tracker.window_batch > 1 and self.do_lc and self.backend is not None and self.mapper is None:
            # windows decoded several at a time WITH the loop-closure backend: its turn comes after every window of the batch, exactly
            # where the one-window loop gives it (the backend then looks at the keyframes tracked so far, not at the ones still waiting)
            def hook(t0, t1):
                if t1 > 10:
                    if self.freeze_counter > 0:
                        did, _ = self.backend.run(t1=t1)
                        if did:
                            self.freeze_counter = 0
                            done.append(t1)
                    else:
                        self.freeze_counter += 1
        run_backend, viz_idx, submap_idx = self.tracker.run(tstamp, last_frame=last_frame, after_window=hook)
        if hook is not None and done:
            lc_did = True
        if run_backend and not last_frame and self.do_lc and self.backend is not None:
            if self.freeze_counter > 0:
                lc_did, updates = self.backend.run()
                if lc_did:
                    self.freeze_counter = 0
                    if self.mapper is not None and getattr(self.mapper, "initialized", False):      # hi2.py:124-131
                        with torch.enable_grad():
                            updated, idx = self.mapper.gaussain_update(updates)
                        if idx:
                            self.keyframes.set_poses_at(idx, updated["poses"].float().cpu().numpy())
                            ds = self.downsample_ratio
                            for j, k in enumerate(idx):
                                self.keyframes.submap_ds[k // 5, k % 5] = updated["pointmaps"][j, ::ds, ::ds]
                            self.keyframes.submap_ds[:submap_idx + 1, -1] = self.keyframes.submap_ds[1:submap_idx + 2, 0]
            else:
                self.freeze_counter += 1
        if viz_idx is not None and self.mapper is not None:
            self.call_gs(viz_idx, submap_idx, self.gs_iter_num, intrinsics_ds)
        return viz_idx, submap_idx, lc_did

    @torch.no_grad()
    def run_buffered(self, frames_u8, intrinsics, t_start=0, lookahead=16, mark_tail=True, on_frame=None, pipeline=False):
        """Buffered-stream driver: the same per-frame `run()` sequence as demo_s.py:151-160 over frames_u8 [n,3,H,W] (time
        stamps t_start..), but in overlap mode the next `lookahead` tested frames (every `skip`-th, plus the always-kept
        second-last / last frame when mark_tail) go through the encoder as ONE batch and their keyframe decisions are taken
        on the device (MotionFilter.prefetch) -- identical keyframes and features, latency lookahead*skip frames.
        pipeline=True: the encoder pass + decision chain of chunk c+1 run on a side stream BESIDE the per-frame loop (tracking windows) of
        chunk c -- the chain of c+1 starts from the last keyframe chunk c is known to yield -- and only their read-back waits."""
        n = frames_u8.shape[0]
        f = self.filterx
        overlap_mode = not (f.kf_every > 0)
        chunk = max(1, int(lookahead)) * max(1, int(f.skip))

        def tested(c0, c1, first_ever):
            idx, forced = [], []
            for i in range(c0, c1):
                t = t_start + i
                tail = mark_tail and i >= n - 2
                if t % f.skip == 0 or tail or (first_ever and i == c0):
                    idx.append(i)
                    forced.append(tail)
            return idx, forced

        def select(idx, forced):
            return frames_u8[idx[0]:idx[-1] + 1:f.skip] if (not any(forced) and len(idx) > 1 and idx[-1] - idx[0] == f.skip * (len(idx) - 1)) \
                else frames_u8[torch.as_tensor(idx, device=frames_u8.device)]

        pipe = bool(pipeline) and overlap_mode
        if pipe and self._ahead_stream is None:
            self._ahead_stream = torch.cuda.Stream()
        pending = None                         # (handle, base) of a chunk whose look-ahead pass is already running
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            if overlap_mode:
                if pending is not None:
                    kept, last_feat = f.prefetch_collect(pending[0], base=pending[1])
                    pending = None
                else:
                    idx, forced = tested(c0, c1, self.keyframes.counter.value == 0)
                    kept, last_feat = 0, None
                    if idx and pipe:
                        kept, last_feat = f.prefetch_collect(f.prefetch_launch(select(idx, forced), [t_start + i for i in idx], forced))
                    elif idx:
                        f.prefetch(select(idx, forced), [t_start + i for i in idx], forced)
                if pipe and c1 < n:
                    idx2, forced2 = tested(c1, min(n, c1 + chunk), False)
                    if idx2:
                        base2 = self.keyframes.counter.value + kept
                        if last_feat is None and self.keyframes.counter.value > 0:
                            last_feat = self.keyframes.feat_slice(self.keyframes.counter.value - 1, self.keyframes.counter.value)[0]
                        if last_feat is not None:
                            pending = (f.prefetch_launch(select(idx2, forced2), [t_start + i for i in idx2], forced2, feat_last=last_feat,
                                                         stream=self._ahead_stream), base2)
            for i in range(c0, c1):
                out = self.run(t_start + i, frames_u8[i:i + 1], intrinsics, frames_u8[i:i + 1], intrinsics,
                               second_last_frame=mark_tail and i == n - 2, last_frame=mark_tail and i == n - 1)
                if on_frame is not None:
                    on_frame(t_start + i, out)

    @torch.no_grad()
    def run_stream(self, items, lookahead=1, on_frame=None):
        """The look-ahead of `run_buffered` over an ITERATOR of per-frame items (tstamp, image, intrinsics, image_ds, intrinsics_ds,
        second_last_frame, last_frame) -- demo.py's loop over `stream.mono_stream` (demo_s.py:151-160).  `lookahead * skip` items are
        held back; in overlap mode their tested frames (every `skip`-th, plus the always-kept first / second-last / last) go through
        the encoder as ONE batch with the keyframe decisions taken on the device, then every held item goes through `run()` in order.
        With `Tracking.frontend.window_batch` > 1 the keyframes found this way are tracked `window_batch` windows at a time.  Keyframes,
        poses, depths and edges are those of the frame-by-frame loop (bit-identical); the price is the latency of the held frames."""
        f = self.filterx
        overlap_mode = not (f.kf_every > 0)
        chunk = max(1, int(lookahead)) * max(1, int(f.skip))
        held = []

        def flush():
            if overlap_mode and lookahead > 1:
                idx = [k for k, it in enumerate(held) if int(it[0]) % f.skip == 0 or it[5] or it[6] or (self.keyframes.counter.value == 0 and k == 0)]
                if idx:
                    f.prefetch(torch.cat([held[k][3][:1] for k in idx], 0), [int(held[k][0]) for k in idx], [bool(held[k][5] or held[k][6]) for k in idx])
            for it in held:
                out = self.run(it[0], it[1], it[2], it[3], it[4], second_last_frame=bool(it[5]), last_frame=bool(it[6]))
                if on_frame is not None:
                    on_frame(it[0], out)
            held.clear()

        for it in items:
            held.append(it)
            if len(held) >= chunk:
                flush()
        flush()

    @torch.no_grad()
    def terminate(self, add_kf=False, gap=30, finalize_iters=None, gaussian_retrain=False, retrain_iters=10000):
        """hi2.py:152-229.  With the Gaussian mapper attached (`self.mapper`) the extra views go to `mapper.add_new_view`, the mapper
        finalises (`GSMapper.finalize`: a global BA of `finalize_iters` iterations, default the configured position_lr_max_steps as the
        reference's `max_steps`; 0 skips it) and its poses are written back (hi2.py:214-216); `gaussian_retrain` rebuilds the map from all
        keyframes first (hi2.py:155-170).  With add_kf (demo_s.py:171 passes True) every pair of consecutive
        keyframes more than `gap` frames apart gets ONE extra view at the middle frame: the kept full-resolution frame is resized
        to the tracking resolution (bilinear, align_corners=False, hi2.py:198) and relocalised against the earlier keyframe by a
        2-view inference (TrackFrontend.predict, track_frontend.py:102-162).  The reference hands these views to
        `mapper.add_new_view`; tracking-only, they are returned: list of dicts (tstamp, pose [7] c2w, depth [H,W],
        pointmap [h,w,3], conf [h,w], submap).  Returns (keyframes.pose as numpy [buffer,7], new views)."""
        kf = self.keyframes
        last = kf.counter.value
        views = []
        if gaussian_retrain and self.mapper is not None and last > 1:        # hi2.py:155-170: a fresh map from every keyframe's pointmap
            n = last - 1
            pms = kf.submap_ds[:(n + 4) // 5, :-1].flatten(0, 1)[:n]
            with torch.enable_grad():
                self.mapper.gaussian_reinit(kf.image[:n], pms, iteration_total=retrain_iters)
        if add_kf:
            if not self.images:
                raise RuntimeError("terminate(add_kf=True) needs the frames: set slam.keep_images = True before run()")
            ts = kf.tstamp[:max(last - 1, 0)]
            H, W = kf.ht, kf.wd
            for i in range(len(ts) - 1):
                t0, t1 = float(ts[i]), float(ts[i + 1])
                if t1 - t0 > gap:
                    interval = (t1 - t0) // 2
                    t_new = int(t0 + interval)
                    img = self.images[t_new]
                    img = img[0] if img.dim() == 4 else img
                    if tuple(img.shape[-2:]) != (H, W):
                        f = torch.nn.functional.interpolate(img[None].float(), size=(H, W), mode="bilinear", align_corners=False)[0]
                        img = f.round().clamp(0, 255).to(torch.uint8)
                    pose, depth, pm, conf = self.tracker.predict(img.to(self.device), kf.image[i], kf.pose[i], kf.depth[i],
                                                                 kf.submap_ds[i // 5, i % 5])
                    views.append({"tstamp": t_new, "pose": pose, "depth": depth, "pointmap": pm, "conf": conf, "submap": i // 5, "image": img})
        if self.mapper is not None and getattr(self.mapper, "initialized", False):
            with torch.enable_grad():
                for v in views:                                           # hi2.py:203-211
                    self.mapper.add_new_view(v["image"], torch.as_tensor(v["pose"]), v["depth"], v["tstamp"], v["submap"])
                poses = self.mapper.finalize(finalize_iters)
            n = min(max(last - 1, 0), poses.shape[0])
            if n:
                kf.set_poses_at(range(n), poses[:n].cpu().numpy())
        return kf.pose.numpy().copy(), views

    def trajectory(self):
        """(tstamps [t], poses [t,7] c2w (t, q_xyzw)) of the tracked keyframes (demo_s.py:97-100)."""
        t = self.keyframes.counter.value - 1
        if self.tracked_only:
            t = min(t, self.tracker.t1)
        return self.keyframes.tstamp[:t].numpy().copy(), self.keyframes.pose[:t].numpy().copy()

    def save_trajectory(self, path, tstamps_full=None):
        ts, poses = self.trajectory()
        if tstamps_full is not None:
            ts = np.asarray(tstamps_full)[ts.astype(int)]
        np.savetxt(path, np.concatenate([ts.reshape(-1, 1), poses], axis=1), fmt="%.4f %.7f %.7f %.7f %.7f %.7f %.7f %.7f")
