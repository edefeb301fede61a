#!/usr/bin/env python3
"""Per-shape timing of every GEMM launch of one bench step (eager, HIP events on the launch stream).
usage: python tools/gemm_shapes.py [window_batch]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_frames
from cut3r_slam_amd import _lib, dist as cdist
from cut3r_slam_amd.config import production_config
from cut3r_slam_amd.model import Cut3rModel
from cut3r_slam_amd.slam import Cut3rSlam
from cut3r_slam_amd.weights import synth_state_dict

WB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
cfg = production_config()
model = Cut3rModel(cfg, synth_state_dict(cfg, 0), dev, minimal=True)
model.use_graphs = False
os.environ["CUT3R_DUAL_STREAM"] = "0"
conf = {"Tracking": {"motion_filter": {"thresh": 0.9, "skip": 5, "kf_every": 10}, "frontend": {"iteration": 0}}}
slam = Cut3rSlam(model, conf, (384, 512), buffer=32 + 10 * WB, device=dev)
runner = cdist.ShardedTracker(slam, 1, 0, wb=WB, pipelined=False)
frames = synth_frames(runner.frames_needed(2, 10, 5), 384, 512, dev)
intr = torch.tensor([256.0, 338.8, 255.8, 191.7])
t = 0
while not slam.keyframes.is_initialized:
    slam.run(t, frames[t:t + 1], intr, frames[t:t + 1], intr)
    t += 1
t = runner.step(frames, t, 10, 5, intr)
torch.cuda.synchronize()

lib = _lib.load()
raw = lib.cut3r_gemm_f16
rec = []


def wrapped(dref, stream):
    d = dref._obj
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = raw(dref, stream)
    e.record()
    rec.append(((d.M, d.N, d.K, max(d.batch, 1), d.conv_k, d.conv_stride, d.shuf, d.tile, d.stages), s, e))
    return rc


lib.cut3r_gemm_f16 = wrapped
t = runner.step(frames, t, 10, 5, intr)
torch.cuda.synchronize()
lib.cut3r_gemm_f16 = raw
agg = collections.defaultdict(lambda: [0, 0.0])
for key, s, e in rec:
    a = agg[key]
    a[0] += 1
    a[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
print(f"window_batch {WB}: {len(rec)} GEMM launches, {tot:.2f} ms per step")
print(f"{'M':>7} {'N':>6} {'K':>6} {'b':>3} cv st sh tile stg {'n':>5} {'ms':>8} {'us/launch':>9} {'TF/s':>7} {'tiles128':>8}")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, b, ck, cs, sh, tile, stg = key
    fl = 2.0 * M * N * K * b * n
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * b
    print(f"{M:7d} {N:6d} {K:6d} {b:3d} {ck:2d} {cs:2d} {sh:2d} {tile:4d} {stg:3d} {n:5d} {ms:8.3f} {1e3 * ms / n:9.1f} {fl / ms / 1e9:7.1f} {tiles:8d}")
