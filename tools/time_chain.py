#!/usr/bin/env python3
"""Host+device cost of the sequential part (chaining + graph update) per window vs the batched network inference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_frames
from cut3r_slam_amd.config import production_config
from cut3r_slam_amd.model import Cut3rModel
from cut3r_slam_amd.slam import Cut3rSlam
from cut3r_slam_amd.weights import synth_state_dict

dev = "cuda:0"
cfg = production_config()
model = Cut3rModel(cfg, synth_state_dict(cfg, 0), dev, minimal=True)
WB = 4
conf = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "kf_every": 10}, "frontend": {"iteration": 0, "window_batch": WB}}}
slam = Cut3rSlam(model, conf, (384, 512), buffer=200, device=dev)
frames = synth_frames(1500, 384, 512, dev)
intr = torch.tensor([256.0, 338.8, 255.8, 191.7])
tr = slam.tracker
t_inf, t_chain, n = [], [], 0
orig_tb = tr.track_batch


def timed_track_batch(ranges):
    kf = slam.keyframes
    torch.cuda.synchronize(); a = time.perf_counter()
    tr.window_features(ranges[0][0], ranges[-1][1])
    feats = torch.stack([tr.window_features(x, y) for x, y in ranges], 0)
    res = model.decode_windows(feats, kf.ht, kf.wd)
    torch.cuda.synchronize(); b = time.perf_counter()
    V = 6
    for j, (x, y) in enumerate(ranges):
        sl = slice(j * V, (j + 1) * V)
        tr.track(x, y, outputs=(res["pts3d_in_self_view"][sl], res["conf_self"][sl], res["camera_pose"][sl]))
        tr.t1 = y
    torch.cuda.synchronize(); c = time.perf_counter()
    t_inf.append(b - a); t_chain.append(c - b)


tr.track_batch = timed_track_batch
for t in range(1400):
    slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
print(f"batches: {len(t_inf)}  inference per window: {1e3*sum(t_inf[2:])/len(t_inf[2:])/WB:.2f} ms   chaining+graph per window: {1e3*sum(t_chain[2:])/len(t_chain[2:])/WB:.2f} ms")
print("edges:", len(slam.graph._ii), "keyframes tracked:", tr.t1)
