#!/usr/bin/env python3
"""The loop-closure-on leg of bench.py with synchronising timers around the parts of a closure (detect, NMS, re-inference, optimise, rewrite).  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import H, W, KF_EVERY, synth_frames
from cut3r_slam_amd import synth, ops
from cut3r_slam_amd.config import production_config
from cut3r_slam_amd.model import Cut3rModel
from cut3r_slam_amd.slam import Cut3rSlam
dev = "cuda:0"
cfg = production_config()
sd = synth.loop_state_dict(cfg, seed=0, enc_residual_gain=0.1, depth_relief=0.02)
model = Cut3rModel(cfg, sd, dev, minimal=True)
config = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "skip_blur": False, "kf_every": KF_EVERY}, "frontend": {"iteration": 2000, "window_batch": int(os.environ.get("LC_WB", "1"))}}}
intr = torch.tensor([600.0 * W / 1200.0, 600.0 * H / 680.0, 599.5 * W / 1200.0, 339.5 * H / 680.0])
WB = int(os.environ.get('LC_WB', '1'))
warm, n = max(160, (10 * WB + 8) * KF_EVERY), 1000
frames = synth_frames(warm + n, H, W, dev, seed=0)
slam = Cut3rSlam(model, config, (H, W), buffer=(warm + n) // KF_EVERY + 16, device=dev)
T = {}
def wrap(obj, name, key):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T.setdefault(key, []).append(time.perf_counter() - t)
        return r
    setattr(obj, name, w)
be = slam.backend
wrap(be, "run", "backend.run"); wrap(be, "track", "track (6-view re-inference)"); wrap(be, "nms", "nms"); wrap(slam.graph, "detect_loop", "detect_loop")
wrap(be, "loop_closure_init", "loop_closure_init"); wrap(be, "loop_closure", "loop_closure"); wrap(be, "_rewrite", "_rewrite")
wrap(ops, "lc_optimize", "ops.lc_optimize"); wrap(ops, "lc_optimize_terms", "ops.lc_optimize_terms")
for t in range(warm):
    slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
for k in T: T[k] = []
tic = time.perf_counter()
for t in range(warm, warm + n):
    slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
torch.cuda.synchronize()
el = time.perf_counter() - tic
print(f"{n/el:.1f} frames/s, total {1e3*el:.0f} ms")
for k, v in T.items():
    if v: print(f"{k:32s} calls {len(v):3d} total {1e3*sum(v):8.1f} ms avg {1e3*sum(v)/len(v):7.2f} ms max {1e3*max(v):7.2f}")
