#!/usr/bin/env python3
"""cProfile of the sequential part (chaining + covisibility-graph update) over many windows, network outputs precomputed."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_frames
from cut3r_slam_amd import dist as cdist
from cut3r_slam_amd.config import production_config
from cut3r_slam_amd.model import Cut3rModel
from cut3r_slam_amd.slam import Cut3rSlam
from cut3r_slam_amd.weights import synth_state_dict

dev = "cuda:0"
cfg = production_config()
model = Cut3rModel(cfg, synth_state_dict(cfg, 0), dev, minimal=True)
WB, STEPS = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 6
conf = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "kf_every": 10}, "frontend": {"iteration": 0}}}
slam = Cut3rSlam(model, conf, (384, 512), buffer=7 + 5 * WB * STEPS + 16, device=dev)
runner = cdist.ShardedTracker(slam, 1, 0, wb=WB, pipelined=False)
frames = synth_frames(runner.frames_needed(STEPS, 10, 5), 384, 512, dev)
intr = torch.tensor([256.0, 338.8, 255.8, 191.7])
t = 0
while not slam.keyframes.is_initialized:
    slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
    t += 1
# collect network outputs of every step first, replay afterwards under the profiler
pend = []
orig_replay = runner._replay
runner._replay = lambda p: pend.append(p)
for _ in range(STEPS):
    t = runner.step(frames, t, 10, 5, intr)
torch.cuda.synchronize()
runner._replay = orig_replay
pr = cProfile.Profile()
tic = time.perf_counter()
pr.enable()
for p in pend:
    orig_replay(p)
torch.cuda.synchronize()
pr.disable()
el = time.perf_counter() - tic
print(f"{WB * STEPS} windows replayed in {1e3 * el:.1f} ms -> {1e3 * el / (WB * STEPS):.3f} ms per window; edges {len(slam.graph._ii)}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
