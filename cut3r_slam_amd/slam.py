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
    def __init__(self, model, config=None, image_size=(384, 512), buffer=512, device="cuda:0"):
        self.model = model
        self.config = config or DEFAULT_CONFIG
        self.device = device
        self.verbose = False
        self.output_dir = None
        self.use_gt = False
        self.downsample_ratio = 2
        self.keyframes = KeyFrame(self.config, image_size, buffer, self.downsample_ratio, device,
                                  feat_dim=model.cfg.enc_embed_dim, patch=model.cfg.patch_size)
        self.graph = FactorGraph(self.keyframes, device=device, max_factors=48)
        self.filterx = MotionFilter(model, self.keyframes, self.config["Tracking"]["motion_filter"], device)
        self.tracker = TrackFrontend(self, self.keyframes, self.config["Tracking"]["frontend"], device)
        self.backend = TrackBackend(self, self.keyframes, self.config["Tracking"]["frontend"], device)
        self.do_lc = self.config["Tracking"]["frontend"].get("iteration", 0) > 0
        self.freeze_counter = 0

    @torch.no_grad()
    def run(self, tstamp, image, intrinsics, image_ds, intrinsics_ds, second_last_frame=False, last_frame=False):
        """hi2.py:101-133 without the GS mapper: image_ds [1,3,H,W] uint8 at tracking resolution."""
        self.filterx.kfFilter(tstamp, image_ds, intrinsics=intrinsics_ds, second_last_frame=second_last_frame,
                              last_frame=last_frame)
        run_backend, viz_idx, submap_idx = self.tracker.run(tstamp, last_frame=last_frame)
        lc_did = False
        if run_backend and not last_frame and self.do_lc and self.backend is not None:
            if self.freeze_counter > 0:
                lc_did, _ = self.backend.run()
                if lc_did:
                    self.freeze_counter = 0
            else:
                self.freeze_counter += 1
        return viz_idx, submap_idx, lc_did

    def trajectory(self):
        """(tstamps [t], poses [t,7] c2w (t, q_xyzw)) of the tracked keyframes (demo_s.py:97-100)."""
        t = self.keyframes.counter.value - 1
        return self.keyframes.tstamp[:t].numpy().copy(), self.keyframes.pose[:t].numpy().copy()

    def save_trajectory(self, path, tstamps_full=None):
        ts, poses = self.trajectory()
        if tstamps_full is not None:
            ts = np.asarray(tstamps_full)[ts.astype(int)]
        np.savetxt(path, np.concatenate([ts.reshape(-1, 1), poses], axis=1), fmt="%.4f %.7f %.7f %.7f %.7f %.7f %.7f %.7f")
