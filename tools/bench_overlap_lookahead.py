#!/usr/bin/env python3
"""Overlap mode (kf_every = -1, skip 5) over a content-driven stream with growing look-ahead of the keyframe test and window batch: frames/s and
bit-equality of keyframes / poses with the first configuration.  usage: bench_overlap_lookahead.py [frames=2800]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import H, W
from cut3r_slam_amd import synth
from cut3r_slam_amd.config import production_config
from cut3r_slam_amd.model import Cut3rModel
from cut3r_slam_amd.slam import Cut3rSlam
dev = "cuda:0"
cfg = production_config()
model = Cut3rModel(cfg, synth.tracking_state_dict(cfg, seed=0, enc_residual_gain=0.1, depth_relief=0.02), dev, minimal=True)
intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2800
warm = 1600
frames = synth.slideshow_stream(warm + N, H, W, hold=10, seed=0, device=dev)
ref = None
for la, wb, pipe in [(16, 1, False), (56, 4, False), (140, 14, False), (280, 28, False), (280, 28, True), (140, 28, True), (560, 28, True)]:
    config = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "skip_blur": False, "kf_every": -1},
                           "frontend": {"iteration": 0, "window_batch": wb}}}
    slam = Cut3rSlam(model, config, (H, W), buffer=(warm + N) // 10 + 16, device=dev)
    slam.run_buffered(frames[:warm], intr, mark_tail=False, lookahead=la, pipeline=pipe)
    torch.cuda.synchronize()
    k0, w0 = slam.keyframes.counter.value, slam.tracker.t1
    tic = time.perf_counter()
    slam.run_buffered(frames[warm:warm + N], intr, t_start=warm, mark_tail=False, lookahead=la, pipeline=pipe)
    torch.cuda.synchronize()
    el = time.perf_counter() - tic
    k = slam.tracker.t1
    pose = slam.keyframes.pose[:k].clone()
    ts = slam.keyframes.tstamp[:slam.keyframes.counter.value].clone()
    if ref is None:
        ref = (pose, ts)
    kk = min(k, ref[0].shape[0])
    same_kf = bool((ts[:min(len(ts), len(ref[1]))] == ref[1][:min(len(ts), len(ref[1]))]).all())
    dpose = float((pose[:kk] - ref[0][:kk]).abs().max())
    print(f"lookahead {la:4d} tested frames, window_batch {wb:3d}, pipeline {int(pipe)}: {N / el:8.1f} frames/s ({1e3 * el / N:.3f} ms/frame), keyframes {slam.keyframes.counter.value - k0}, "
          f"tracked to {k}, same keyframes as the first run: {same_kf}, max |pose diff| over the common prefix {dpose:.2e}", flush=True)
